"""column-sliced SpMM against the L2 row-gather SpMM on the BASELINE configs[4] graph (bench.c5_graph: nnz 1.99 M)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from beta_recsys_amd import _lib
from beta_recsys_amd.lightgcn import (_csr_from_coo, _slice_rows, sliced_graph_device, sliced_graph_host,
                                      spread_bank_conflicts)
from oracle import lightgcn_numpy as olg

import bench

D = 64
adj = bench.c5_graph()  # 988 k unique edges, nnz 1 985 746
N = adj.shape[0]
lib = _lib.load()
dev = torch.device("cuda:0")
st = _lib.stream_ptr(dev)
r, c, v = (torch.from_numpy(x) for x in (adj.row.astype(np.int64), adj.col.astype(np.int64), adj.data.astype(np.float32)))
rp, cc, vv, _ = _csr_from_coo(r, c, v, N, dev)
nnz = adj.nnz
x = torch.randn(N, D, device=dev)
y = torch.zeros(N, D, device=dev)
acc = torch.zeros(N, D, device=dev)


def timed(fn, n=100):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


sl = _slice_rows(rp, cc, vv, N, nnz)
csr = _lib.Csr(rp.data_ptr(), cc.data_ptr(), vv.data_ptr(), None, N, nnz, sl.data_ptr())
print(f"gather SpMM   nnz {nnz}: {timed(lambda: _lib.check(lib.hiprec_spmm_csr(ctypes.byref(csr), None, 1.0, _lib.ptr(x), _lib.ptr(y), _lib.ptr(acc), D, st))):6.1f} us")
W = lib.hiprec_sliced_width(N, D)
cap = lib.hiprec_sliced_row_cap(N, D)
for n_groups, factor, spread in ((16, False, False), (16, True, False), (16, True, True)):
    host = sliced_graph_host(rp.cpu().numpy(), cc.cpu().numpy(), vv.cpu().numpy(), None, n_groups, cap, factor=factor)
    if spread:
        print("bank-conflict ways before / after the slot permutation: %.2f / %.2f" % spread_bank_conflicts(host, n_groups))
    sc, hold = sliced_graph_device(host, N, n_groups, cap, dev)
    xs, ys, accs = (torch.zeros(N * D, device=dev) for _ in range(3))
    _lib.check(lib.hiprec_to_sliced(_lib.ptr(x), N, D, W, None, _lib.ptr(xs), st))
    t = timed(lambda: _lib.check(lib.hiprec_spmm_sliced(ctypes.byref(sc), None, 1.0, _lib.ptr(xs), _lib.ptr(ys), _lib.ptr(accs), 1, D, W, st)))
    print(f"sliced SpMM W {W} factored {'col_scale' in host} groups {n_groups} subs/group {host['subs_per_group']} chunks {host['n_chunks']} slots {host['n_slots']}: {t:6.1f} us"
          "", flush=True)
    if "col_scale" in host:  # LDS conflict share: every slot reads the zero row (one address: broadcast) / a conflict-free pattern
        z = torch.full((host["n_slots"],), N, dtype=torch.int16, device=dev)
        t = timed(lambda: _lib.check(lib.hiprec_spmm_sliced(ctypes.byref(sc), _lib.ptr(z), 1.0, _lib.ptr(xs), _lib.ptr(ys), _lib.ptr(accs), 1, D, W, st)))
        print(f"   all slots -> zero row (broadcast reads): {t:6.1f} us")
        seq = (torch.arange(host["n_slots"], device=dev) // 16 % 9000).to(torch.int16)  # lane-constant columns, quads differ
        t = timed(lambda: _lib.check(lib.hiprec_spmm_sliced(ctypes.byref(sc), _lib.ptr(seq), 1.0, _lib.ptr(xs), _lib.ptr(ys), _lib.ptr(accs), 1, D, W, st)))
        print(f"   lane-constant consecutive rows (conflict-free): {t:6.1f} us")
