"""Per-rank cost of the row-sharded step at the configs[3] shard size (world size 1: the all-to-alls are
self-copies, so this is everything except the xGMI transfer time)."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
from beta_recsys_amd.sharded import ShardedMFEngine
U, I, D, B = 1_250_000, 125_000, 128, 65536
for sgd_mode in ("rows", "dense"):
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer="sgd", lr=0.05, batch_size=B,
                         loss="bpr", routing="padded", sgd_mode=sgd_mode), "system": {"run_dir": "/tmp/x"}}
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ShardedMFEngine(cfg)
    g = torch.Generator().manual_seed(1)
    n_steps = 30
    users = torch.randint(0, U, (n_steps * B,), generator=g).cuda()
    p = 1.0 / torch.arange(1, I + 1, dtype=torch.float64)
    pos = torch.multinomial(p / p.sum(), n_steps * B, True, generator=g).cuda()
    neg = torch.randint(0, I, (n_steps * B,), generator=g).cuda()
    def run(a, b):
        for k in range(a, b):
            sl = slice(k * B, (k + 1) * B)
            eng.train_single_batch((users[sl], pos[sl], neg[sl]), sync=False)
    run(0, 5); torch.cuda.synchronize(); t0 = time.perf_counter(); run(5, n_steps); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n_steps - 5)
    eng.k.check_status()
    print(f"sharded step, one rank's configs[3] share, sgd_mode={sgd_mode}: {dt * 1e6:.0f} us/step = {B / dt / 1e6:.0f} M triples/s")
    del eng
    torch.cuda.empty_cache()
# the epoch-planned path (plan_epoch once per epoch, then 2 collectives + 5 launches per step)
import beta_recsys_amd as hp
cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer="sgd", lr=0.05, batch_size=B,
                     loss="bpr", sgd_mode="rows"), "system": {"run_dir": "/tmp/x"}}
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    eng = ShardedMFEngine(cfg)
loader = hp.DeviceTripleBatcher(users, pos, neg, B)
torch.cuda.synchronize(); t0 = time.perf_counter()
plan = eng.plan_epoch(loader)
torch.cuda.synchronize(); t_plan = time.perf_counter() - t0
t0 = time.perf_counter(); plan = eng.plan_epoch(loader); torch.cuda.synchronize(); t_plan2 = time.perf_counter() - t0
eng.run_planned_epoch(plan)
torch.cuda.synchronize(); t0 = time.perf_counter()
eng.run_planned_epoch(plan, sync=False)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / plan["S"]
print(f"planned sharded epoch: plan {t_plan * 1e3:.1f} ms first / {t_plan2 * 1e3:.1f} ms warm for {plan['S']} steps "
      f"({t_plan2 / plan['S'] * 1e6:.0f} us/step), step {dt * 1e6:.0f} us = {B / dt / 1e6:.0f} M triples/s; "
      f"slots/step {sum(plan['n_slots']) / plan['S']:.0f} of {2 * B} references")
dist.destroy_process_group()
