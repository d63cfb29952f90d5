"""where the staging of a big-batch epoch spends its time (steady state, HIP events)"""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beta_recsys_amd as hp
from beta_recsys_amd import _lib
from beta_recsys_amd.mf import batch_row_ownership, sort_within_batches

U, I, D, B, S = 1_250_000, 125_000, 128, 65536, 50
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
n = S * B
users = torch.randint(0, U, (n,), generator=g).to(dev)
pos = torch.randint(0, I, (n,), generator=g).to(dev)
neg = torch.randint(0, I, (n,), generator=g).to(dev)
lib = _lib.load()


def timed(name, fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:34s} {e0.elapsed_time(e1) / reps * 1e3 / S:8.1f} us per step ({e0.elapsed_time(e1) / reps:7.2f} ms per epoch)")
    return out


def permutation():
    perm = torch.empty(n, dtype=torch.int64, device=dev)
    _lib.check(lib.hiprec_random_permutation(_lib.ptr(perm), n, 12345, _lib.stream_ptr(dev)))
    return perm


perm = timed("feistel permutation", permutation)
order = timed("sort_within_batches (argsort)", lambda: sort_within_batches(perm, pos, B, I))
su, sp, sn = timed("3 gathers", lambda: (users[order], pos[order], neg[order]))
timed("ownership (hash tables)", lambda: batch_row_ownership(su, sp, sn, B, U, I))
key = (torch.arange(n, device=dev) // B) * I + pos[perm]
timed("  key build", lambda: (torch.arange(n, device=dev) // B) * I + pos[perm])
timed("  argsort int64", lambda: torch.argsort(key))
k32 = key.to(torch.int32)
timed("  argsort int32", lambda: torch.argsort(k32))
timed("  sort int32 (values+idx)", lambda: torch.sort(k32))
