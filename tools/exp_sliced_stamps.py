"""In-kernel timestamps of one pass of the column-sliced SpMM on the C5 graph (training stream: 40 % of the slots
dropped).  Needs the debug library: tools/build_debug_lib.sh, then HIPREC_LIB=libhiprec_debug.so python
tools/exp_sliced_stamps.py [lane_slots].  Prints, for a first-wave and a last-wave thread of two workgroups, the ticks
(s_memtime: the shader clock here, ~2.3 GHz) between phase boundaries, next to the HIP-event time of the pass."""
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
W, cap, n_groups = lib.hiprec_sliced_width(N, D), lib.hiprec_sliced_row_cap(N, D), 16
S = int(sys.argv[1]) if len(sys.argv) > 1 else None
host = sliced_graph_host(rp.cpu().numpy(), cc.cpu().numpy(), vv.cpu().numpy(), None, n_groups, cap, factor=True, lane_slots=S)
spread_bank_conflicts(host, n_groups)
sc, hold = sliced_graph_device(host, N, n_groups, cap, dev)
xs, ys, accs = (torch.zeros(N * D, device=dev) for _ in range(3))
_lib.check(lib.hiprec_to_sliced(_lib.ptr(x), N, D, W, None, _lib.ptr(xs), st))
live = torch.from_numpy(host["col16"].astype(np.int64)).to(dev)
drop = torch.rand(live.numel(), device=dev) >= 0.6
train = torch.where(drop, torch.full_like(live, N), live).to(torch.int16)


def one():
    _lib.check(lib.hiprec_spmm_sliced(ctypes.byref(sc), _lib.ptr(train), 1.0, _lib.ptr(xs), _lib.ptr(ys), _lib.ptr(accs), 1, D, W, st))


for _ in range(20):
    one()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(200):
    one()
e1.record(); torch.cuda.synchronize()
print(f"lane_slots {host['lane_slots']}: {e0.elapsed_time(e1) / 200 * 1e3:.2f} us per pass (HIP events, back to back)")
dbg = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 64)()
dbg.hiprec_debug_sliced_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert dbg.hiprec_debug_sliced_stamps(out) == 0
names = {0: "entry", 1: "first descriptors requested", 2: "slice stored to LDS", 3: "barrier", 4: "loop starts", 5: "chunks done",
         6: "barrier (spill / empty rows)", 15: "end"}
for k, who in enumerate(("block 0 wave 0", "block 0 wave 15", "block 131 wave 0", "block 131 wave 15")):
    s_ = list(out[k * 16:(k + 1) * 16])
    print(f"{who}: total {s_[15] - s_[0]} cycles")
    prev = s_[0]
    for i, nm in names.items():
        if s_[i]:
            print(f"   {nm:28s} +{s_[i] - prev:6d}  (at {s_[i] - s_[0]})")
            prev = s_[i]
