"""Host time of one `_enqueue_step` call against the GPU time of the step it enqueues (LightGCN, NeuMF at the BASELINE
shapes): python tools/exp_host_enqueue.py.  The loop is timed twice: while the queue is being drained (host + GPU
together = what a short bench window sees) and with a synchronize only at the end (the host alone, if it is the faster)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")


def measure(name, eng, batches, n=400):
    for b in batches[:20]:
        eng._enqueue_step(*b) if isinstance(b, tuple) and name == "ncf" else eng._enqueue_step(b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for k in range(n):
        b = batches[k % len(batches)]
        eng._enqueue_step(*b) if name == "ncf" else eng._enqueue_step(b)
    t_host = time.perf_counter() - t0
    e1.record()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{name}: host loop {t_host / n * 1e6:7.1f} us per step, GPU events {e0.elapsed_time(e1) / n * 1e3:7.1f}, "
          f"wall incl. drain {t_all / n * 1e6:7.1f}")


import numpy as np
import beta_recsys_amd as hp
from bench import synth_triples, c5_graph, U, I, D

g = torch.Generator().manual_seed(1)
adj = c5_graph()
idx = torch.from_numpy(np.vstack((adj.row, adj.col)).astype(np.int64))
norm = torch.sparse_coo_tensor(idx, torch.from_numpy(adj.data), torch.Size(adj.shape))
cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, layer_size=[D] * 3, keep_pro=0.6, regs=[1e-5], device_str=str(dev),
                     optimizer="adam", lr=0.05, batch_size=1024, norm_adj=norm, dropout_rng="device"),
       "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
torch.manual_seed(2020)
with contextlib.redirect_stdout(io.StringIO()):
    eng = hp.LightGCNEngine(cfg)
users, pos, neg = (t.to(dev) for t in synth_triples(64 * 1024, seed=100))
batches = [(users[k * 1024:(k + 1) * 1024], pos[k * 1024:(k + 1) * 1024], neg[k * 1024:(k + 1) * 1024]) for k in range(64)]
measure("lightgcn", eng, batches)

cfg = {"model": dict(n_users=U, n_items=I, emb_dim=32, dropout=0.0, device_str=str(dev), optimizer="adam", lr=1e-3,
                     batch_size=4096, model="ncf_end", mlp_config={"n_layers": 3}, gmf_config={}),
       "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
with contextlib.redirect_stdout(io.StringIO()):
    ncf = hp.NeuMFEngine(cfg)
u = torch.randint(0, U, (16 * 4096,), generator=g).to(dev)
i = torch.randint(0, I, (16 * 4096,), generator=g).to(dev)
r = (torch.rand(16 * 4096, generator=g) < 0.2).float().to(dev)
batches = [(u[k * 4096:(k + 1) * 4096], i[k * 4096:(k + 1) * 4096], r[k * 4096:(k + 1) * 4096]) for k in range(16)]
measure("ncf", ncf, batches)
