"""Bounded replay (csrc/lazy_opt.hip) under a stopwatch: N rows of dim 128 take one Adam step at t = 1 and then lag for
GAP steps; time (a) the catch-up that brings their weights up to date and (b) the flush of the same state.
    python tools/exp_lazy_gap.py [GAP ...]        (HIPREC_LIB=libhiprec_<variant>.so for an A/B)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from beta_recsys_amd import _lib  # noqa: E402
from test_lazy_opt_gpu import Lazy, new_stats  # noqa: E402

lib, dev = _lib.load(), torch.device("cuda:0")
U, I, D, lr = 120_000, 8, 128, 0.05
P = (U + I) * (D + 1) + 1
st = _lib.stream_ptr(dev)
for gap in [int(a) for a in sys.argv[1:]] or [40, 150, 400, 2000]:
    out = []
    for what in ("catch-up", "flush"):
        gen = torch.Generator(device="cuda").manual_seed(3)
        w = torch.randn(P, device=dev, generator=gen) * 0.1
        g, m, v = torch.zeros_like(w), torch.zeros_like(w), torch.zeros_like(w)
        stats = new_stats(lib, _lib, dev)
        lazy = Lazy(lib, _lib, w, g, m, v, U, I, D, 1, lr, cap=max(4096, gap + 8))
        rows = torch.arange(U, dtype=torch.int64, device=dev)
        none = torch.zeros(0, dtype=torch.int64, device=dev)
        g[: U * D] = torch.randn(U * D, device=dev, generator=gen) * 0.01
        _lib.check(lib.hiprec_stats_advance_step(_lib.ptr(stats), st))
        lazy.update(stats, rows, None, None, None)          # step 1: every row gets its moments
        for _ in range(gap):                                # the clock runs on, nobody touches the rows
            _lib.check(lib.hiprec_stats_advance_step(_lib.ptr(stats), st))
            lazy.update(stats, none, None, None, None)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if what == "catch-up":
            lazy.catchup(stats, rows, None, None, None)
        else:
            lazy.flush(stats)
        b.record()
        torch.cuda.synchronize()
        out.append(f"{what} {a.elapsed_time(b) * 1e3:9.1f} us")
    print(f"[{os.environ.get('HIPREC_LIB', 'libhiprec.so')}] {U} rows x dim {D}, gap {gap:5d} steps: " + ", ".join(out), flush=True)
