"""Where a pass of the column-sliced SpMM spends its time on the C5 graph (nnz 1 985 746): the kernel's experiment bits
(HIPREC_SLICED_EXP, read per launch, only in a library built by tools/build_debug_lib.sh and selected with
HIPREC_LIB=libhiprec_debug.so: 1 slice-major block map, 4 no output stores, 16 no slice fill; results are wrong with
4 / 16: timing only) and the debug edge streams (40 % of the slots -> the zero row as in a training step; every slot ->
the zero row; lane-constant consecutive rows).  profiles/r03_experiments.md 39."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from beta_recsys_amd import _lib
from beta_recsys_amd.lightgcn import _csr_from_coo, sliced_graph_device, sliced_graph_host, spread_bank_conflicts

D = 64
adj = bench.c5_graph()
N = adj.shape[0]
lib = _lib.load()
dev = torch.device("cuda:0")
st = _lib.stream_ptr(dev)
r, c, v = (torch.from_numpy(x) for x in (adj.row.astype(np.int64), adj.col.astype(np.int64), adj.data.astype(np.float32)))
rp, cc, vv, _ = _csr_from_coo(r, c, v, N, dev)
x = torch.randn(N, D, device=dev)


def timed(fn, n=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


W = lib.hiprec_sliced_width(N, D)
cap = lib.hiprec_sliced_row_cap(N, D)
n_groups = 16
S = int(sys.argv[1]) if len(sys.argv) > 1 else None   # slots per lane (None: the host's choice)
host = sliced_graph_host(rp.cpu().numpy(), cc.cpu().numpy(), vv.cpu().numpy(), None, n_groups, cap, factor=True, lane_slots=S)
print("lane_slots", host["lane_slots"])
print("bank-conflict ways before / after the slot permutation: %.2f / %.2f" % spread_bank_conflicts(host, n_groups))
sc, hold = sliced_graph_device(host, N, n_groups, cap, dev)
xs, ys, accs = (torch.zeros(N * D, device=dev) for _ in range(3))
_lib.check(lib.hiprec_to_sliced(_lib.ptr(x), N, D, W, None, _lib.ptr(xs), st))
print(f"chunks {host['n_chunks']} slots {host['n_slots']} subs/group {host['subs_per_group']} row_cap {cap}")
# the training step's stream: 40 % of the live slots point at the zero row
live = torch.from_numpy(host["col16"].astype(np.int64)).to(dev)
drop = torch.rand(live.numel(), device=dev) >= 0.6
train = torch.where(drop, torch.full_like(live, N), live).to(torch.int16)
zero = torch.full((host["n_slots"],), N, dtype=torch.int16, device=dev)
seq = (torch.arange(host["n_slots"], device=dev) // 16 % 9000).to(torch.int16)
streams = {"graph": None, "training (40 % dropped)": train, "all zero row": zero, "conflict-free": seq}
for exp in (0, 0, 1, 4, 16, 4 | 16, 32, 4 | 16 | 32):
    os.environ["HIPREC_SLICED_EXP"] = str(exp)
    row = []
    for name, e in streams.items():
        t = timed(lambda: _lib.check(lib.hiprec_spmm_sliced(ctypes.byref(sc), _lib.ptr(e), 1.0, _lib.ptr(xs), _lib.ptr(ys),
                                                            _lib.ptr(accs), 1, D, W, st)))
        row.append(f"{name} {t:5.1f}")
    print(f"exp {exp:2d}: " + " | ".join(row), flush=True)
