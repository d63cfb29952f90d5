import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beta_recsys_amd import _lib
dev = torch.device("cuda:0")
eng = bench.make_engine(dev, "sgd"); eng._setup()
B = bench.B
users, pos, neg = (t.to(dev) for t in bench.synth_triples(B, 1))
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "exp_libs", "ts.so"))
fn = lib.hiprec_mf_bpr_grad; fn.restype = ctypes.c_int; fn.argtypes = _lib.SIGNATURES["hiprec_mf_bpr_grad"][1]
scratch = torch.zeros(65536 + 4096 * 64, dtype=torch.uint8, device=dev)
m = eng.model; w, g = m.tables(), m.tables(eng._g_flat); st = _lib.stream_ptr(dev)
args = (ctypes.byref(w), ctypes.byref(g), _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), None, B, 1.0/B, 0.0,
        _lib.ptr(eng._stats), _lib.ptr(scratch), scratch.numel(), st)
for _ in range(5): assert fn(*args) == 0
torch.cuda.synchronize()
ts = scratch.cpu().numpy()[65536:].view(np.uint64).reshape(4096, 8).astype(np.int64)
wall = ts[:, 6:8]; ts = ts[:, :6]
w0 = wall[:, 0].min()
st_ = (wall[:, 0] - w0) * 10; en_ = (wall[:, 1] - w0) * 10
print('wall-clock (ns): start p10/p50/p90/max', np.percentile(st_, [10, 50, 90, 100]), ' end p10/p50/p90/max', np.percentile(en_, [10, 50, 90, 100]))
hist = np.histogram(st_, bins=10)[0]; print('start histogram over kernel span:', hist)
print('wave wall duration ns median', np.median(en_ - st_))
d = np.diff(ts, axis=1)
print('per-wave stage durations (median cycles):', np.median(d, axis=0), ' p90:', np.percentile(d, 90, axis=0))
