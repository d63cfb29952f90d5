import ctypes, sys, os, glob
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from beta_recsys_amd import _lib
dev = torch.device("cuda:0")
eng = bench.make_engine(dev, "sgd"); eng._setup()
B = bench.B
users, pos, neg = (t.to(dev) for t in bench.synth_triples(B, 1))
for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "exp_libs", "*.so"))):
    lib = ctypes.CDLL(path)
    fn = lib.hiprec_mf_bpr_grad; fn.restype = ctypes.c_int; fn.argtypes = _lib.SIGNATURES["hiprec_mf_bpr_grad"][1]
    m = eng.model; w, g = m.tables(), m.tables(eng._g_flat); st = _lib.stream_ptr(dev)
    args = (ctypes.byref(w), ctypes.byref(g), _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), None, B, 1.0/B, 0.0,
            _lib.ptr(eng._stats), _lib.ptr(eng._scratch), eng._scratch.numel(), st)
    for _ in range(10): assert fn(*args) == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(200): fn(*args)
    b.record(); torch.cuda.synchronize()
    print("%-50s %.2f us" % (os.path.basename(path), a.elapsed_time(b) / 200 * 1e3))
    eng._g_flat.zero_()
