import sys, os, time, cProfile, pstats, contextlib, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import beta_recsys_amd as hp
dev = torch.device("cuda:0")
U, I, B = bench.U, bench.I, bench.B
cfg = {"model": dict(n_users=U, n_items=I, emb_dim=32, dropout=0.0, device_str="cuda:0", optimizer="adam", lr=1e-3,
                     batch_size=B, model="ncf_end", mlp_config={"n_layers": 3}, gmf_config={}),
       "system": {"run_dir": "/tmp/x"}}
with contextlib.redirect_stdout(io.StringIO()):
    eng = hp.NeuMFEngine(cfg)
users = torch.randint(0, U, (B,), device=dev); items = torch.randint(0, I, (B,), device=dev); r = (torch.rand(B, device=dev) < 0.2).float()
for _ in range(10): eng._enqueue_step(users, items, r)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): eng._enqueue_step(users, items, r)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue per step %.1f us, total per step %.1f us" % ((t1 - t0) * 1e4, (t2 - t0) * 1e4))
pr = cProfile.Profile(); pr.enable()
for _ in range(100): eng._enqueue_step(users, items, r)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(12); print(s.getvalue()[:2500])
