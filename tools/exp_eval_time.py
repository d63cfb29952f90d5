"""Time hiprec_rank_metrics at ML-1M evaluation sizes."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from beta_recsys_amd import eval as hev  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
out = {}
for name, U, C, pos in (("leave_one_out_6040x101", 6040, 101, None), ("full_catalogue_6040x3706", 6040, 3706, 0.01)):
    users = torch.arange(U).repeat_interleave(C).to(dev)
    if pos is None:
        ratings = torch.zeros(U, C)
        ratings[:, 0] = 1
        ratings = ratings.reshape(-1)
    else:
        ratings = (torch.rand(U * C, generator=gen) < pos).float()
    ratings = ratings.to(dev)
    scores = torch.rand(U * C, generator=gen).to(dev)
    ks = [5, 10, 20]
    for _ in range(2):
        hev.rank_metrics(users, ratings, scores, ks)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        n, table = hev.rank_metrics(users, ratings, scores, ks)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / reps * 1e3
    # kernel alone (inputs grouped): HIP events around the library call
    lib = hev._lib.load()
    order, seg = hev.group_by_user(users)
    ws = torch.empty(lib.hiprec_rank_metrics_workspace_bytes(U, 3) // 8, dtype=torch.float64, device=dev)
    o = torch.empty(13, dtype=torch.float64, device=dev)
    import ctypes
    kk = (ctypes.c_int32 * 3)(*ks)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.hiprec_rank_metrics(hev._lib.ptr(seg), U, hev._lib.ptr(scores), hev._lib.ptr(ratings), kk, 3,
                                hev._lib.ptr(ws), ws.numel() * 8, hev._lib.ptr(o), hev._lib.stream_ptr(dev))
    e1.record()
    torch.cuda.synchronize()
    kern_ms = e0.elapsed_time(e1) / reps
    out[name] = dict(rows=U * C, end_to_end_ms=round(gpu_ms, 3), kernels_ms=round(kern_ms, 3),
                     rows_per_s=round(U * C / (kern_ms * 1e-3)), ndcg_at_10=float(table[1, 2]))
print(json.dumps(out))
