"""How much of NeuMF's grouped weight-gradient launch is the GEMMs and how much the tables' optimizer sweep riding in
it?  hiprec_ncf_grad (the same launches WITHOUT the sweep blocks and without the tail) against hiprec_ncf_step, under
rocprofv3 --kernel-trace --stats; with HIPREC_LIB=libhiprec_test.so, HIPREC_GEMM_EXP = 1 (split-K partial products
stored plainly instead of added atomically), 2 (GEMM tiles return at once), 4 (column sums do) ablate parts of the
grouped launch (timing only: the results are wrong).   python tools/exp_ncf_wgrad_split.py [grad|step] [emb_dim]"""
import contextlib, ctypes, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beta_recsys_amd as hp
from beta_recsys_amd import _lib

mode = sys.argv[1] if len(sys.argv) > 1 else "grad"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 32
U, I, B = 6040, 3706, 4096
cfg = {"model": dict(n_users=U, n_items=I, emb_dim=E, dropout=0.0, device_str="cuda:0", optimizer="adam", lr=1e-3,
                     batch_size=B, model="ncf_end", mlp_config={"n_layers": 3}, gmf_config={}),
       "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
torch.manual_seed(2020)
with contextlib.redirect_stdout(io.StringIO()):
    eng = hp.NeuMFEngine(cfg)
g = torch.Generator().manual_seed(1)
batches = [(torch.randint(0, U, (B,), generator=g).cuda(), torch.randint(0, I, (B,), generator=g).cuda(),
            (torch.rand(B, generator=g) < 0.2).float().cuda()) for _ in range(8)]
lib = eng._setup()
m = eng.model
st = _lib.stream_ptr(m.flat.device)
for k in range(300):
    u, i, r = batches[k % 8]
    if mode == "step":
        eng._enqueue_step(u, i, r)
    else:
        plan = m.plan(B, eng._g_flat)
        _lib.check(lib.hiprec_ncf_grad(ctypes.byref(plan), _lib.ptr(u), _lib.ptr(i), _lib.ptr(r), B, 1.0 / B,
                                       _lib.ptr(eng._stats), _lib.ptr(eng._scratch), eng._scratch.numel(), st))
torch.cuda.synchronize()
print("done", mode, E)
if _lib.LIB_PATH.endswith("debug.so"):   # tools/build_debug_lib.sh: in-kernel timestamps of two weight-gradient tiles
    dl = ctypes.CDLL(_lib.LIB_PATH)
    out = (ctypes.c_ulonglong * 32)()
    dl.hiprec_debug_gemm_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    assert dl.hiprec_debug_gemm_stamps(out) == 0
    names = {0: "entry", 1: "offsets, two stages of loads issued", 2: "step 0 operands in LDS", 3: "step 0 multiplied",
             4: "step 1 in LDS", 5: "step 1 multiplied", 6: "step 2 in LDS", 7: "step 2 multiplied", 8: "step 3 in LDS",
             9: "step 3 multiplied", 12: "epilogue issued"}
    for blk in range(2):
        st = list(out[blk * 16:(blk + 1) * 16])
        print(f"block {'0' if blk == 0 else '200'}: total {st[12] - st[0]} ticks")
        prev = st[0]
        for k in names:
            if st[k]:
                print(f"   {names[k]:40s} +{st[k] - prev:7d}   (at {st[k] - st[0]})")
                prev = st[k]
