"""In-kernel timestamps of the fused NCF launch (forward + head + input-gradient chain + scatter) at the BASELINE
configs[2] shape.  Needs the debug library: tools/build_debug_lib.sh, then HIPREC_LIB=libhiprec_debug.so python
tools/exp_ncf_stamps.py [emb_dim].  Prints, for two blocks, the cycles between phase boundaries (thread 0)."""
import contextlib, ctypes, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beta_recsys_amd as hp
from beta_recsys_amd import _lib

E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
U, I, B = 6040, 3706, 4096
dev = torch.device("cuda:0")
cfg = {"model": dict(n_users=U, n_items=I, emb_dim=E, dropout=0.0, device_str=str(dev), optimizer="adam", lr=1e-3,
                     batch_size=B, model="ncf_end", mlp_config={"n_layers": 3}, gmf_config={}),
       "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    eng = hp.NeuMFEngine(cfg)
g = torch.Generator().manual_seed(1)
users = torch.randint(0, U, (60 * B,), generator=g).to(dev)
items = torch.randint(0, I, (60 * B,), generator=g).to(dev)
ratings = (torch.rand(60 * B, generator=g) < 0.2).float().to(dev)
for k in range(60):
    sl = slice(k * B, (k + 1) * B)
    eng._enqueue_step(users[sl], items[sl], ratings[sl])
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)  # the debug entry point is not part of the declared ABI
out = (ctypes.c_ulonglong * 48)()
lib.hiprec_debug_ncf_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert lib.hiprec_debug_ncf_stamps(out) == 0
names = {0: "start", 1: "indices in LDS", 2: "gathered rows in LDS", 3: "layer 1", 4: "layer 2", 5: "layer 3", 8: "head",
         9: "d w_out, partials", 17: " (barrier)", 18: " (chain 3 gemm)", 19: " (next layer's loads issued)",
         20: " (chain 3 epilogue)", 10: "chain layer 3", 11: "chain layer 2", 12: "chain layer 1", 13: "partials, d w_out rows", 14: "(non-chained: act stores)", 16: "last pass of the tower-input atomics, end"}
for blk in range(2):
    st = list(out[blk * 24:(blk + 1) * 24])
    print(f"block {'0' if blk == 0 else '131'}: total {st[16] - st[0]} ticks")
    prev = st[0]
    for k in names:
        if st[k]:
            print(f"   {names[k]:24s} +{st[k] - prev:7d}   (at {st[k] - st[0]})")
            prev = st[k]
