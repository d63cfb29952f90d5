"""Where does the replicated engine's per-step time go at world size 1?  (host-bound path)"""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
import beta_recsys_amd.replicated as rep
from beta_recsys_amd.mf import sort_within_batches
U, I, D, B = 6040, 3706, 64, 4096
cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer="adam", lr=0.05, batch_size=B, loss="bpr"), "system": {"run_dir": "/tmp/x"}}
with contextlib.redirect_stdout(io.StringIO()):
    eng = rep.ReplicatedMFEngine(cfg)
eng.presorted = True
n = 600 * B
g = torch.Generator().manual_seed(0)
users = torch.randint(0, U, (n,), generator=g).cuda(); pos = torch.randint(0, I, (n,), generator=g).cuda(); neg = torch.randint(0, I, (n,), generator=g).cuda()
perm = sort_within_batches(torch.randperm(n, device="cuda"), pos, B, I)
users, pos, neg = users[perm], pos[perm], neg[perm]
def run(k0, k1):
    for k in range(k0, k1):
        sl = slice(k * B, (k + 1) * B)
        eng.enqueue_presorted(users[sl], pos[sl], neg[sl])
def timeit(label):
    run(0, 100); torch.cuda.synchronize(); t0 = time.perf_counter(); run(100, 600); t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label}: {dt / 500 * 1e6:.1f} us/step (host enqueue {t_host / 500 * 1e6:.1f})")
timeit("full step")
orig = rep.allreduce_sum_
rep.allreduce_sum_ = lambda buf, group=None: buf
timeit("without the all-reduce call")
rep.allreduce_sum_ = orig
t0 = time.perf_counter()
for k in range(500):
    sl = slice(k * B, (k + 1) * B); a, b, c = users[sl], pos[sl], neg[sl]
print(f"three tensor slices: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us")
buf = eng._g_ext
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(500): dist.all_reduce(buf)
t_host = time.perf_counter() - t0; torch.cuda.synchronize()
print(f"all_reduce alone: host {t_host / 500 * 1e6:.1f} us, total {(time.perf_counter() - t0) / 500 * 1e6:.1f} us")
dist.destroy_process_group()
