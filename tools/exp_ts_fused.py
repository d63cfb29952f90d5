import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beta_recsys_amd import _lib
import beta_recsys_amd as hp
dev = torch.device("cuda:0")
eng = bench.make_engine(dev, "sgd"); eng._setup()
B = bench.B; nsteps = 6
users, pos, neg = (t.to(dev) for t in bench.synth_triples(B * nsteps, 1))
prep = eng.prepare_epoch(hp.DeviceTripleBatcher(users, pos, neg, B))
pu, pp, pn, _, _ = prep
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "exp_libs", "ts_fused.so"))
fn = lib.hiprec_mf_bpr_epoch_sgd_fused; fn.restype = ctypes.c_int; fn.argtypes = _lib.SIGNATURES["hiprec_mf_bpr_epoch_sgd_fused"][1]
m = eng.model
big = 65536 + 4096 * 64
w_alt = torch.empty_like(m.flat); g = [torch.zeros_like(m.flat) for _ in range(3)]
sc = [torch.zeros(big, dtype=torch.uint8, device=dev) for _ in range(2)]
w_arr = (ctypes.c_void_p * 2)(m.flat.data_ptr(), w_alt.data_ptr()); g_arr = (ctypes.c_void_p * 3)(*(t.data_ptr() for t in g)); s_arr = (ctypes.c_void_p * 2)(*(t.data_ptr() for t in sc))
final = ctypes.c_int32(-1)
for rep in range(3):
    rc = fn(w_arr, g_arr, s_arr, m.n_users, m.n_items, m.emb_dim, _lib.ptr(pu), _lib.ptr(pp), _lib.ptr(pn), B * nsteps, B, 0.0, 0.05, _lib.ptr(eng._stats), ctypes.byref(final), _lib.stream_ptr(dev)); assert rc == 0
torch.cuda.synchronize()
for which in (0, 1):
    ts = sc[which].cpu().numpy()[65536:].view(np.uint64).reshape(4096, 8).astype(np.int64)
    wall = ts[:, 6:8]; cyc = ts[:, :6]
    w0 = wall[:, 0].min(); st_ = (wall[:, 0] - w0) * 10; en_ = (wall[:, 1] - w0) * 10
    print("scratch", which, "start p50/max", np.percentile(st_, [50, 100]), "end p10/p50/p90/max", np.percentile(en_, [10, 50, 90, 100]))
    d = np.diff(cyc, axis=1)
    print("   stage cycles median [gb-ready, idx, rows+reduce, math+atomics, merge/flush/publish]:", np.median(d, axis=0), "p90", np.percentile(d, 90, axis=0))
