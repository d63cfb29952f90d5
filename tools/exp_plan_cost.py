"""where does ShardedMFEngine.plan_epoch spend its GPU time? (configs[3] size at world 1: 10 M x 1 M tables are not
needed for the plan -- only the ids -- so the tables here are small)   run under rocprofv3 --kernel-trace --stats"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
import beta_recsys_amd as hp
from beta_recsys_amd.sharded import ShardedMFEngine

Uc, Ic, Dc, Bc, S = 10_000_000, 1_000_000, 8, 65536, 50
cfg = {"model": dict(n_users=Uc, n_items=Ic, emb_dim=Dc, device_str="cuda:0", optimizer="sgd", lr=0.05, batch_size=Bc,
                     loss="bpr", sgd_mode="rows", shard_init="local"), "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
with contextlib.redirect_stdout(io.StringIO()):
    eng = ShardedMFEngine(cfg)
g = torch.Generator().manual_seed(5)
n = S * Bc
users = torch.randint(0, Uc, (n,), generator=g).cuda()
pz = 1.0 / torch.arange(1, Ic + 1, dtype=torch.float64)
pos = torch.multinomial(pz / pz.sum(), n, True, generator=g).cuda()
neg = torch.randint(0, Ic, (n,), generator=g).cuda()
loader = hp.DeviceTripleBatcher(users, pos, neg, Bc)
for _ in range(2):
    eng.plan_epoch(loader)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5):
    eng.plan_epoch(loader)
torch.cuda.synchronize()
print(f"plan_epoch: {(time.time() - t0) / 5 * 1e3:.2f} ms per {S}-step epoch of {Bc} triples = {(time.time() - t0) / 5 / S * 1e6:.1f} us per step", file=sys.stderr)
dist.destroy_process_group()
