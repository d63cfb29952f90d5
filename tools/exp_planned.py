"""The epoch-planned row-sharded path at world size 1 (everything except the xGMI transfer time): plan cost, steps
alone through the C driver and through the torch.distributed loop, SGD and Adam.  SIZE=shard|full (configs[3] one
rank's share / whole), run plain or under rocprofv3 --kernel-trace --stats."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
torch.cuda.set_device(0)
import beta_recsys_amd as hp
from beta_recsys_amd.sharded import ShardedMFEngine
full = os.environ.get("SIZE", "shard") == "full"
U, I, D, B = (10_000_000, 1_000_000, 128, 65536) if full else (1_250_000, 125_000, 128, 65536)
S = int(os.environ.get("STEPS", "30"))
g = torch.Generator().manual_seed(1)
users = torch.randint(0, U, (S * B,), generator=g).cuda()
p = 1.0 / torch.arange(1, I + 1, dtype=torch.float64)
pos = torch.randperm(I, generator=g)[torch.multinomial(p / p.sum(), S * B, True, generator=g)].cuda()
neg = torch.randint(0, I, (S * B,), generator=g).cuda()
loader = hp.DeviceTripleBatcher(users, pos, neg, B)
for opt, driver in [c.split(":") for c in os.environ.get("CASES", "sgd:c,sgd:torch,adam:c").split(",")]:
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer=opt, lr=0.05, batch_size=B,
                         loss="bpr", sgd_mode="rows", shard_init="local", step_driver=driver,
                         dense_opt=os.environ.get("DENSE_OPT", "auto"), shard_sgd=os.environ.get("SHARD_SGD", "pull")),
           "system": {"run_dir": "/tmp/x"}}
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ShardedMFEngine(cfg)
    if os.environ.get("LAZY_FLUSH", "epoch") != "epoch":   # timing experiment: rows stay lagging between epochs
        eng.flush_lazy_every_epoch = False
    plan = eng.plan_epoch(loader)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        plan = eng.plan_epoch(loader)
    torch.cuda.synchronize(); t_plan = (time.perf_counter() - t0) / 3
    eng.run_planned_epoch(plan)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    E = int(os.environ.get("EPOCHS", "3"))
    for _ in range(E):
        eng.run_planned_epoch(plan, sync=False)
    e1.record(); torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / (E * plan["S"])
    eng.k.check_status()
    lazy = "" if opt == "sgd" else (" lazy" if eng._lazy is not None else " sweep")
    print(f"[{'full' if full else 'shard'}] {opt}{lazy} / {driver}: plan {t_plan * 1e3:.2f} ms per {plan['S']} steps "
          f"({t_plan / plan['S'] * 1e6:.0f} us/step), step {dt * 1e6:.1f} us = {B / dt / 1e6:.0f} M triples/s; "
          f"slots/step {sum(plan['n_slots']) / plan['S']:.0f} of {2 * B} references, cap {plan['cap']}", file=sys.stderr)
    del eng, plan
    torch.cuda.empty_cache()
dist.destroy_process_group()
