"""Experiment: how much of the BPR grad kernel's time is atomic contention on popular items?"""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beta_recsys_amd import _lib

def time_grad(eng, users, pos, neg, n=100):
    lib = eng._setup(); m = eng.model; Bn = users.numel()
    w, g = m.tables(), m.tables(eng._g_flat); st = _lib.stream_ptr(m.flat.device)
    args = (ctypes.byref(w), ctypes.byref(g), _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), None, Bn,
            1.0 / Bn, 0.0, _lib.ptr(eng._stats), _lib.ptr(eng._scratch), eng._scratch.numel(), st)
    for _ in range(10): _lib.check(lib.hiprec_mf_bpr_grad(*args))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): _lib.check(lib.hiprec_mf_bpr_grad(*args))
    b.record(); torch.cuda.synchronize()
    eng._g_flat.zero_()
    return a.elapsed_time(b) / n * 1e3

dev = torch.device("cuda:0")
eng = bench.make_engine(dev, "sgd")
B = bench.B
users, pos, neg = (t.to(dev) for t in bench.synth_triples(B, 1))
print("zipf pos        : %.2f us" % time_grad(eng, users, pos, neg))
upos = torch.randint(0, bench.I, (B,), device=dev)
print("uniform pos     : %.2f us" % time_grad(eng, users, upos, neg))
print("all-distinct    : %.2f us" % time_grad(eng, torch.arange(B, device=dev) % bench.U, torch.arange(B, device=dev) % bench.I, (torch.arange(B, device=dev) + 1234) % bench.I))
print("single hot item : %.2f us" % time_grad(eng, users, torch.zeros_like(pos), neg))
sp, order = torch.sort(pos)
print("zipf sorted     : %.2f us" % time_grad(eng, users[order], sp, neg[order]))
for nb in (256, 1024, 16384):
    u2, p2, n2 = (t.to(dev) for t in bench.synth_triples(nb, 2))
    print("zipf B=%5d    : %.2f us" % (nb, time_grad(eng, u2, p2, n2)))
