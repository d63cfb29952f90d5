"""timing breakdown of the owned-rows step (HIPREC_OWNED_DBG variants are wrong on purpose: timing only)"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beta_recsys_amd as hp

full = "--full" in sys.argv
form = next((a.split("=")[1] for a in sys.argv if a.startswith("--form=")), "owned")   # owned (pull) | owned_atomic
U, I, D, B = (10_000_000, 1_000_000, 128, 65536) if full else (1_250_000, 125_000, 128, 65536)
dev = torch.device("cuda:0")
cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer="sgd", lr=0.05, batch_size=B,
                     loss="bpr", sgd_mode=form), "system": {"run_dir": "/tmp/x"}}
with contextlib.redirect_stdout(io.StringIO()):
    eng = hp.MFEngine(cfg)
g = torch.Generator().manual_seed(5)
n = 20 * B
users = torch.randint(0, U, (n,), generator=g).to(dev)
pz = 1.0 / torch.arange(1, I + 1, dtype=torch.float64)
pos = torch.randperm(I, generator=g)[torch.multinomial(pz / pz.sum(), n, True, generator=g)].to(dev)
neg = torch.randint(0, I, (n,), generator=g).to(dev)
loader = hp.DeviceTripleBatcher(users, pos, neg, B, shuffle=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
prep = eng.prepare_epoch(loader)
torch.cuda.synchronize(); print("staging ms (20 steps)", (time.perf_counter() - t0) * 1e3)
own = prep.own[0]
print("shared fraction user/pos/neg:", [(own[r] >= 0).float().mean().item() for r in range(3)])
if form == "owned":
    print("records short / long / contributions per batch:", prep.own[2][:3, :3].tolist())
for rep in range(2):
    eng.run_prepared_epoch(prep, sync=False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); eng.run_prepared_epoch(prep, sync=False); e1.record(); torch.cuda.synchronize()
print(form, "full" if full else "shard", "dbg", os.environ.get("HIPREC_OWNED_DBG", "0"), "us/step", e0.elapsed_time(e1) * 1e3 / 20)
