"""bench.py — BPR-MF training throughput on MI355X (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--optimizer sgd|adam|rmsprop]

A "step" is one pass of the hot path over one batch of B = 4096 synthetic (user, pos, neg) triples:
gather -> score -> BPR gradient -> scatter into the dense gradient -> optimizer update (exactly
the work of MFEngine.train_single_batch in the reference, beta_rec/models/mf.py:92-119).  Inputs
(triples + permutation) are resident in HBM when the timed region starts; the K timed steps are
enqueued by the library's epoch driver and bracketed by barrier + synchronize on both sides.
Rank 0 prints ONE JSON line.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# BASELINE.json configs[1] / SURVEY.md §8 C2
U, I, D, B = 6040, 3706, 64, 4096
LR = 0.05
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec peak)


def algorithmic_bytes_per_triple(dim):
    """indices 3*8 B + 3 row reads + 3 row writes of (dim+1) fp32."""
    return 24 + 24 * (dim + 1)


def synth_triples(n, seed):
    """MovieLens-1M-shaped synthetic: uniform users, Zipf(1.0) positives over a seeded permutation
    of the items, uniform negatives."""
    g = torch.Generator().manual_seed(seed)
    users = torch.randint(0, U, (n,), generator=g)
    p = 1.0 / torch.arange(1, I + 1, dtype=torch.float64)
    pos = torch.randperm(I, generator=g)[torch.multinomial(p / p.sum(), n, True, generator=g)]
    neg = torch.randint(0, I, (n,), generator=g)
    return users, pos, neg


def make_engine(device, optimizer):
    import beta_recsys_amd as hp

    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str=str(device), optimizer=optimizer,
                         lr=LR, batch_size=B, loss="bpr"),
           "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
    torch.manual_seed(2020)
    with contextlib.redirect_stdout(io.StringIO()):
        return hp.MFEngine(cfg)


def stage(eng, batcher):
    """Inputs resident in HBM before the clock starts: triples + this epoch's permutation."""
    prepared = eng.prepare_epoch(batcher)
    torch.cuda.synchronize()
    return prepared


def kernel_timing(eng, prepared, n_launch=200):
    """Average duration of ONE launch of the dominant kernel (BPR grad: gather+score+scatter),
    measured live with HIP events on the stream the kernel is launched on.  It is launched exactly
    as the epoch driver launches it: one batch of the staged epoch, read through perm[]."""
    import ctypes

    from beta_recsys_amd import _lib

    lib = eng._setup()
    m = eng.model
    users, pos, neg, perm, _ = prepared
    w, g = m.tables(), m.tables(eng._g_flat)
    st = _lib.stream_ptr(m.flat.device)
    args = (ctypes.byref(w), ctypes.byref(g), _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg),
            _lib.ptr(perm), B, 1.0 / B, 0.0, _lib.ptr(eng._stats), _lib.ptr(eng._scratch),
            eng._scratch.numel(), st)
    for _ in range(20):
        _lib.check(lib.hiprec_mf_bpr_grad(*args))
    torch.cuda.synchronize()
    # (a) back-to-back launches between ONE event pair: duration + the ~1 us launch-to-launch gap
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n_launch):
        _lib.check(lib.hiprec_mf_bpr_grad(*args))
    b.record()
    torch.cuda.synchronize()
    back_to_back = a.elapsed_time(b) / n_launch * 1e-3
    # (b) one event pair per launch: includes the event packets themselves (upper bound)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(n_launch)]
    for a, b in evs:
        a.record()
        _lib.check(lib.hiprec_mf_bpr_grad(*args))
        b.record()
    torch.cuda.synchronize()
    per = sorted(a.elapsed_time(b) for a, b in evs)  # ms
    eng._g_flat.zero_()
    eng.load_optimizer_state(0)
    return back_to_back, float(per[len(per) // 2]) * 1e-3  # seconds


def cpu_baseline(optimizer="adam", budget_s=12.0):
    """The reference's CPU path (PyTorch ops, dense autograd, torch.optim) on this box's host
    cores, timed on a bounded sample of the same workload: oracle/torch_port.py, kind "port".
    ATen's intra-op threading hurts these small ops on many-core hosts, so a few thread counts are
    tried and the FASTEST is reported (cores = the thread count that won)."""
    from oracle import mf_numpy as onp
    from oracle.torch_port import TorchMFPort

    n_batches = 64
    users, pos, neg = synth_triples(n_batches * B, seed=1)
    batches = [(users[i * B:(i + 1) * B], pos[i * B:(i + 1) * B], neg[i * B:(i + 1) * B])
               for i in range(n_batches)]
    all_threads = torch.get_num_threads()
    candidates = sorted({all_threads, min(32, all_threads), min(8, all_threads)}, reverse=True)
    best = None
    for nt in candidates:
        torch.set_num_threads(nt)
        torch.manual_seed(0)
        port = TorchMFPort(onp.init_params(U, I, D, seed=0), optimizer, LR, "bpr")
        for i in range(3):
            port.step(batches[i])
        steps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s / len(candidates):
            port.step(batches[steps % n_batches])
            steps += 1
        dt = time.perf_counter() - t0
        rate = steps * B / dt
        if best is None or rate > best[0]:
            best = (rate, nt, steps, dt)
    rate, nt, steps, dt = best
    # (ii) end to end through DataLoader(PairwiseNegativeDataset, shuffle=True) as the reference
    # feeds it (data/base_data.py:247-253): per-sample __getitem__ + default_collate dominate
    from torch.utils.data import DataLoader, Dataset

    class _Pairs(Dataset):
        def __getitem__(self, i):
            return users[i], pos[i], neg[i]

        def __len__(self):
            return users.size(0)

    torch.set_num_threads(nt)
    port = TorchMFPort(onp.init_params(U, I, D, seed=0), optimizer, LR, "bpr")
    e2e_steps, t0 = 0, time.perf_counter()
    for batch in DataLoader(_Pairs(), batch_size=B, shuffle=True):
        port.step(batch)
        e2e_steps += 1
        if time.perf_counter() - t0 > 3.0:
            break
    e2e_rate = e2e_steps * B / (time.perf_counter() - t0)
    torch.set_num_threads(all_threads)
    return {"value": rate, "unit": "triples/s", "cores": nt, "kind": "port",
            "end_to_end_dataloader_value": e2e_rate,
            "sample": f"{steps} {optimizer} steps of batch {B} (same C2 workload) in {dt:.1f} s with {nt} ATen "
                      f"threads (best of {candidates}); PyTorch-CPU op sequence of the reference; host has "
                      f"{os.cpu_count()} logical cpus"}


def measured_traffic_bytes(kernel="hiprec::mf_bpr_grad_kernel<1>"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_summary.json: FETCH_SIZE + WRITE_SIZE, KB per dispatch), or None."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    try:
        with open(path) as f:
            k = json.load(f)[kernel]
        return (k["FETCH_SIZE_KB_mean"] + k["WRITE_SIZE_KB_mean"]) * 1024.0
    except Exception:
        return None


def other_workload_traffic(workload, kernel):
    """HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, KB per dispatch) of a kernel of another bench
    workload from the committed rocprofv3 PMC passes (profiles/r01_pmc_other_workloads.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_other_workloads.json")) as f:
            k = json.load(f)[workload][kernel]
        return (k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
    except Exception:
        return None


def bench_ncf(args, device):
    """BASELINE configs[2]: NeuMF (GMF + MLP [128, 64, 32] <=> emb_dim 32, quirk Q9) on the ML-1M
    shape, batch 4096 (user, item, rating) samples with 1 positive : 4 negatives, Adam lr 1e-3."""
    import beta_recsys_amd as hp

    E, L = 32, 3
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=E, dropout=0.0, device_str=str(device),
                         optimizer="adam", lr=1e-3, batch_size=B, model="ncf_end",
                         mlp_config={"n_layers": L}, gmf_config={}),
           "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
    torch.manual_seed(2020)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = hp.NeuMFEngine(cfg)
    n_total = (args.warmup + args.steps) * B
    # the sample stream instance_bce_loader builds (data/base_data.py:182-216, num_negative 4): every
    # positive (Zipf item, rating 1) is followed by 4 negatives of the same user (uniform items, rating
    # 0), then the DataLoader shuffles the samples
    pu, ppos, pneg = synth_triples(n_total // 5 + 1, seed=100)
    g = torch.Generator().manual_seed(101)
    users = pu.repeat_interleave(5)[:n_total]
    items = torch.cat([ppos[:, None], torch.randint(0, I, (ppos.numel(), 4), generator=g)], 1).reshape(-1)[:n_total]
    ratings = torch.tensor([1.0, 0, 0, 0, 0]).repeat(ppos.numel())[:n_total]
    shuffle = torch.randperm(n_total, generator=g)
    users, items, ratings = (t[shuffle].contiguous().to(device) for t in (users, items, ratings))

    def run(lo, n):
        for k in range(n):
            sl = slice(lo + k * B, lo + (k + 1) * B)
            eng._enqueue_step(users[sl], items[sl], ratings[sl])

    run(0, args.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.warmup * B, args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng._sync_stats()
    flops = 3 * 2 * (256 * 128 + 128 * 64 + 64 * 32 + 64)  # fwd + dgrad + wgrad per sample
    out = {"metric": "training interactions/sec (NCF samples)", "value": args.steps * B / dt,
           "unit": "samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "NeuMF (BASELINE configs[2]): 6040 x 3706, emb_dim 32 => tables "
                                  "128/128/32/32, tower 256->128->64->32, head 64->1, batch 4096, adam 1e-3",
                      "last_loss": st.loss},
           "mfma": {"flops_per_sample": flops, "achieved_tflops": args.steps * B * flops / dt / 1e12,
                    "peak_fp32_mfma_tflops": 157.3}}
    print(json.dumps(out), flush=True)


def bench_mf_c4shard(args, device, full=False):
    """BASELINE configs[3] (10M users x 1M items, dim 128) in the HBM-resident regime of the SAME
    gradient kernel the headline runs in its cache-resident regime, batch 65536, exact SGD on the
    touched rows.  `full=False`: one rank's share of the 8-GPU layout (1.25M x 128 and 125k x 128
    rows, 0.7 GB + as much gradient), no exchange timed.  `full=True`: the WHOLE configuration on
    one MI355X -- 5.7 GB of tables + 5.7 GB of dense gradient buffer fit 288 GB of HBM many times
    over, so a single GPU needs no sharding and no all-to-all at all for this size."""
    import ctypes

    import beta_recsys_amd as hp
    from beta_recsys_amd import _lib

    Uc, Ic, Dc, Bc = (10_000_000, 1_000_000, 128, 65536) if full else (1_250_000, 125_000, 128, 65536)
    cfg = {"model": dict(n_users=Uc, n_items=Ic, emb_dim=Dc, device_str=str(device), optimizer="sgd",
                         lr=LR, batch_size=Bc, loss="bpr", sgd_mode="rows"),
           "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
    torch.manual_seed(2020)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = hp.MFEngine(cfg)
    steps, warm = min(args.steps, 200), min(args.warmup, 20)
    n_total = (steps + warm) * Bc
    g = torch.Generator().manual_seed(5)
    users = torch.randint(0, Uc, (n_total,), generator=g).to(device)
    pz = 1.0 / torch.arange(1, Ic + 1, dtype=torch.float64)
    pos = torch.randperm(Ic, generator=g)[torch.multinomial(pz / pz.sum(), n_total, True, generator=g)].to(device)
    neg = torch.randint(0, Ic, (n_total,), generator=g).to(device)
    nw = warm * Bc
    eng.run_prepared_epoch(stage(eng, hp.DeviceTripleBatcher(users[:nw], pos[:nw], neg[:nw], Bc)))
    prepared = stage(eng, hp.DeviceTripleBatcher(users[nw:], pos[nw:], neg[nw:], Bc))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_prepared_epoch(prepared, sync=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.epoch_stats()
    # dominant kernel alone, back to back
    lib = eng._setup()
    m = eng.model
    pu, pp, pn, _, _ = prepared
    w, gt = m.tables(), m.tables(eng._g_flat)
    sp = _lib.stream_ptr(device)
    kargs = (ctypes.byref(w), ctypes.byref(gt), _lib.ptr(pu), _lib.ptr(pp), _lib.ptr(pn), None, Bc,
             1.0 / Bc, 0.0, _lib.ptr(eng._stats), _lib.ptr(eng._scratch), eng._scratch.numel(), sp)
    for _ in range(5):
        _lib.check(lib.hiprec_mf_bpr_grad(*kargs))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(50):
        _lib.check(lib.hiprec_mf_bpr_grad(*kargs))
    b.record()
    torch.cuda.synchronize()
    k_s = a.elapsed_time(b) / 50 * 1e-3
    bpt = algorithmic_bytes_per_triple(Dc)
    out = {"metric": "training interactions/sec (BPR triples)", "value": steps * Bc / dt, "unit": "triples/s",
           "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": dt / steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": ("BPR-MF, BASELINE configs[3] whole on one GPU: 10M x 1M rows, dim 128, batch "
                                   "65536, exact SGD on touched rows" if full else
                                   "BPR-MF, one rank's shard of BASELINE configs[3]: 1.25M x 125k rows, dim 128, "
                                   "batch 65536, exact SGD on touched rows (no exchange timed)"),
                      "last_loss": st.loss},
           "roofline": {"bound": "hbm", "kernel": "mf_bpr_grad_kernel<2>", "achieved": bpt * Bc / k_s / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bpt * Bc / k_s / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": bpt * Bc, "kernel_us": k_s * 1e6,
                        "traffic": None if full else other_workload_traffic("mf-c4shard", "hiprec::mf_bpr_grad_kernel<2>"),
                        "step_frac": steps * Bc / dt * bpt / (HBM_PEAK_GBS * 1e9)}}
    print(json.dumps(out), flush=True)


def bench_lightgcn(args, device):
    """BASELINE configs[4]: LightGCN on an ML-1M-sized graph (~1M interactions, nnz ~2M), 3 layers,
    dim 64, batch 1024 triples, keep_pro 0.6 (device-side edge dropout), Adam lr 0.05."""
    import beta_recsys_amd as hp

    L, Bl = 3, 1024
    rng = np.random.default_rng(0)
    n_edges = 1_000_000
    p = 1.0 / np.arange(1, I + 1) ** 0.9
    eu = rng.integers(0, U, n_edges)
    ei = rng.permutation(I)[rng.choice(I, n_edges, p=p / p.sum())]
    # the reference's norm_adj = D^-1 (A + I) over users + items (data/deprecated_data_base.py:331-353 +
    # utils/common_util.py normalized_adj_single), built once on the host like the reference does
    import scipy.sparse as sp

    n_nodes = U + I
    rows, cols = np.concatenate([eu, ei + U]), np.concatenate([ei + U, eu])
    a = sp.coo_matrix((np.ones(len(rows), dtype=np.float32), (rows, cols)), shape=(n_nodes, n_nodes)).tocsr()
    a.data[:] = 1.0  # duplicate interactions are one edge
    a = a + sp.eye(n_nodes, dtype=np.float32, format="csr")
    adj = sp.diags(1.0 / np.asarray(a.sum(1)).flatten()).dot(a).astype(np.float32).tocoo()
    idx = torch.from_numpy(np.vstack((adj.row, adj.col)).astype(np.int64))
    norm = torch.sparse_coo_tensor(idx, torch.from_numpy(adj.data), torch.Size(adj.shape))
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, layer_size=[D] * L, keep_pro=0.6, regs=[1e-5],
                         device_str=str(device), optimizer="adam", lr=0.05, batch_size=Bl, norm_adj=norm,
                         dropout_rng="device"),
           "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
    torch.manual_seed(2020)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = hp.LightGCNEngine(cfg)
    n_total = (args.warmup + args.steps) * Bl
    users, pos, neg = (t.to(device) for t in synth_triples(n_total, seed=100))

    def run(lo, n):
        for k in range(n):
            sl = slice(lo + k * Bl, lo + (k + 1) * Bl)
            eng._enqueue_step((users[sl], pos[sl], neg[sl]))

    run(0, args.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.warmup * Bl, args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng._sync_stats()
    nnz, N = adj.nnz, U + I
    # 2L SpMMs x [nnz*(4+4) + (N+1)*8 + 2*N*D*4] bytes (+ the keep byte per edge)
    bytes_step = 2 * L * (nnz * 9 + (N + 1) * 8 + 2 * N * D * 4)
    out = {"metric": "training interactions/sec (LightGCN triples)", "value": args.steps * Bl / dt,
           "unit": "triples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"LightGCN (BASELINE configs[4]): 6040 x 3706 graph, nnz {nnz}, 3 layers, "
                                  "dim 64, batch 1024, keep_pro 0.6 (device RNG), adam 0.05",
                      "last_loss": st.loss},
           "roofline": {"bound": "hbm", "algorithmic_bytes_per_step": bytes_step,
                        "achieved": bytes_step / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": bytes_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBS}}
    print(json.dumps(out), flush=True)


def port_baseline(make_port, batches, units_per_step, what, budget_s=9.0):
    """cpu_baseline for the sibling workloads: the reference's ATen op sequence (oracle/torch_port.py,
    pinned on goldens from the real reference) on this box's host cores, bounded sample, best of a few
    intra-op thread counts (cores = the count that won)."""
    all_threads = torch.get_num_threads()
    candidates = sorted({all_threads, min(32, all_threads), min(8, all_threads)}, reverse=True)
    best = None
    for nt in candidates:
        torch.set_num_threads(nt)
        torch.manual_seed(0)
        port = make_port()
        port.step(batches[0])
        steps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s / len(candidates):
            port.step(batches[(steps + 1) % len(batches)])
            steps += 1
        dt = time.perf_counter() - t0
        if best is None or steps / dt > best[0] / best[3]:
            best = (steps, nt, units_per_step, dt)
    torch.set_num_threads(all_threads)
    steps, nt, units, dt = best
    return {"value": steps * units / dt, "unit": "triples/s", "cores": nt, "kind": "port",
            "sample": f"{steps} steps of {what} in {dt:.1f} s with {nt} ATen threads (best of {candidates}); "
                      f"PyTorch-CPU op sequence of the reference; host has {os.cpu_count()} logical cpus"}


def bench_siblings(args, device):
    """SURVEY.md §8f rank 4 siblings at the reference's own default shapes on ML-1M-sized tables:
    pgmf = PairwiseGMF (configs/cmn_default.json: emb_dim 64, batch 1024, adam 1e-4, l2 1e-4, clip 5);
    t2v  = Triple2vec  (configs/triple2vec_default.json: emb_dim 64, n_neg 5, batch 256, adam 5e-4)."""
    import beta_recsys_amd as hp

    run_dir = {"run_dir": "/tmp/hiprec_bench_runs"}
    torch.manual_seed(2020)
    g = torch.Generator().manual_seed(102)
    if args.workload == "pgmf":
        Bs = 1024
        cfg = {"n_users": U, "n_items": I, "emb_dim": D, "regs": [1e-5], "batch_size": Bs, "lr": 1e-4,
               "pretrain_l2_lambda": 1e-4, "grad_clip": 5.0, "neg_count": 4,
               "model": {"device_str": str(device), "optimizer": "adam", "lr": 1e-4}, "system": run_dir}
        with contextlib.redirect_stdout(io.StringIO()):
            eng = hp.PairwiseGMFEngine(cfg)
        n_total = (args.warmup + args.steps) * Bs
        cols = [t.to(device) for t in synth_triples(n_total, seed=100)]
        batch_of = lambda sl: tuple(c[sl] for c in cols)  # noqa: E731
        P = eng.model.flat.numel()
        # per triple: 3 ids + 3 rows read + 3 rows accumulated; per step: clip reads g once (4 P), the
        # Adam sweep reads w, m, v, g and writes w, m, v (28 P)
        bytes_step = Bs * (24 + 24 * D) + 32 * P
        label = (f"PairwiseGMF (cmn_default.json): 6040 x 3706, emb_dim 64, batch {Bs}, adam 1e-4, "
                 "l2 1e-4, grad_clip 5.0")
        unit = "triples/s"
    else:
        Bs, n_neg = 256, 5
        cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, n_neg=n_neg, batch_size=Bs, device_str=str(device),
                             optimizer="adam", lr=5e-4), "system": run_dir}
        with contextlib.redirect_stdout(io.StringIO()):
            eng = hp.Triple2vecEngine(cfg)
        n_total = (args.warmup + args.steps) * Bs
        u, i1, _ = synth_triples(n_total, seed=100)
        _, i2, _ = synth_triples(n_total, seed=101)
        _, nb1, _ = synth_triples(n_total * n_neg, seed=103)   # popularity-weighted, like the alias sampler
        _, nb2, _ = synth_triples(n_total * n_neg, seed=104)
        nu = torch.randint(0, U, (n_total, n_neg), generator=g)
        cols = [t.to(device) for t in (u, i1, i2, nu, nb1.reshape(n_total, n_neg), nb2.reshape(n_total, n_neg))]
        batch_of = lambda sl: tuple(c[sl] for c in cols)  # noqa: E731
        eng.model._alias()
        P = eng.model.n_active()
        # per triple (item_emb2 aliased): 3 + 3 n_neg ids, 3 + 2 n_neg rows read and as many
        # accumulated, 3 + 3 n_neg bias reads + updates; per step the Adam sweep (28 P)
        bytes_step = Bs * ((3 + 3 * n_neg) * 8 + 2 * (3 + 2 * n_neg) * D * 4 + 2 * (3 + 3 * n_neg) * 4) + 28 * P
        label = (f"Triple2vec (triple2vec_default.json): 6040 x 3706, emb_dim 64, n_neg {n_neg}, "
                 f"batch {Bs}, adam 5e-4")
        unit = "triples/s"

    def run(lo, n):  # n steps = one resident "epoch" of n * Bs triples, enqueued by the library's C driver
        eng.enqueue_epoch(*batch_of(slice(lo, lo + n * Bs)))

    run(0, args.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.warmup * Bs, args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng._sync_stats()
    out = {"metric": f"training interactions/sec ({args.workload} triples)", "value": args.steps * Bs / dt,
           "unit": unit, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": label, "last_loss": st.loss},
           "roofline": {"bound": "hbm", "algorithmic_bytes_per_step": bytes_step,
                        "achieved": bytes_step / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": bytes_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                        "note": "whole step (all launches); at these batch sizes the step is launch-latency bound"}}
    if not args.no_cpu_baseline:
        from oracle import torch_port

        w0 = {k: v.detach().cpu().numpy() for k, v in eng.model.state_dict().items()}
        host = [tuple(c[k * Bs:(k + 1) * Bs].cpu() for c in cols) for k in range(8)]
        if args.workload == "pgmf":
            make = lambda: torch_port.TorchPGMFPort(w0, "adam", 1e-4, 1e-4, 5.0)  # noqa: E731
        else:
            make = lambda: torch_port.TorchT2VPort(w0, Bs, "adam", 5e-4)  # noqa: E731
        out["cpu_baseline"] = port_baseline(make, host, Bs, f"batch {Bs} (same workload)")
    print(json.dumps(out), flush=True)


def bench_ngcf(args, device):
    """NGCF at configs/ngcf_default.json (emb 64, three hops of 64, mess_dropout 0.1,
    batch 1024, Adam lr 0.05) on the same ML-1M-sized graph as the LightGCN workload."""
    import beta_recsys_amd as hp
    import scipy.sparse as sp

    L, Bn = 3, 1024
    rng = np.random.default_rng(0)
    n_edges = 1_000_000
    p = 1.0 / np.arange(1, I + 1) ** 0.9
    eu = rng.integers(0, U, n_edges)
    ei = rng.permutation(I)[rng.choice(I, n_edges, p=p / p.sum())]
    n_nodes = U + I
    rows, cols = np.concatenate([eu, ei + U]), np.concatenate([ei + U, eu])
    a = sp.coo_matrix((np.ones(len(rows), dtype=np.float32), (rows, cols)), shape=(n_nodes, n_nodes)).tocsr()
    a.data[:] = 1.0
    a = a + sp.eye(n_nodes, dtype=np.float32, format="csr")
    adj = sp.diags(1.0 / np.asarray(a.sum(1)).flatten()).dot(a).astype(np.float32).tocoo()
    idx = torch.from_numpy(np.vstack((adj.row, adj.col)).astype(np.int64))
    norm = torch.sparse_coo_tensor(idx, torch.from_numpy(adj.data), torch.Size(adj.shape))
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, layer_size=[D] * L, mess_dropout=[0.1] * L, regs=[1e-5],
                         device_str=str(device), optimizer="adam", lr=0.05, batch_size=Bn, norm_adj=norm,
                         dropout_rng="device"),
           "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
    torch.manual_seed(2020)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = hp.NGCFEngine(cfg)
    eng.model.train()
    n_total = (args.warmup + args.steps) * Bn
    users, pos, neg = (t.to(device) for t in synth_triples(n_total, seed=100))

    def run(lo, n):
        for k in range(n):
            sl = slice(lo + k * Bn, lo + (k + 1) * Bn)
            eng._enqueue_step((users[sl], pos[sl], neg[sl]))

    run(0, args.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.warmup * Bn, args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng._sync_stats()
    nnz, N = adj.nnz, U + I
    act = N * D * 4                      # one [N, D] fp32 activation
    spmm = nnz * 8 + (N + 1) * 8 + 2 * act
    # per hop, forward: SpMM; bi_mul (2 in, 1 out); two Linear (2 in, 2 out); act (2 in + keep byte, ego + slice out)
    fwd = spmm + 3 * act + 4 * act + (4 * act + N * D)
    # per hop, backward: act_bwd (d_all + all slices, d_next, keep, 2 pre in; 2 out); six grouped problems
    # (d_sum, d_bi read by dgrad + wgrad + colsum; side, bi_in in; d_side, d_bi_in out); bi_bwd (4 in, 2 out);
    # transposed SpMM (+ accumulate into d_ego: 2 more)
    bwd = (5 * act + N * D + 2 * act) + (6 * act + 2 * act + 2 * act) + 6 * act + (spmm + 2 * act)
    P = eng.model.flat.numel()
    bytes_step = L * (fwd + bwd) + 2 * N * 4 * D * 4 + Bn * (24 + 6 * 4 * D * 4) + 28 * P
    flops = L * 6 * 2 * N * D * D        # two Linear per hop: forward + dgrad + wgrad
    out = {"metric": "training interactions/sec (NGCF triples)", "value": args.steps * Bn / dt,
           "unit": "triples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"NGCF (ngcf_default.json): 6040 x 3706 graph, nnz {nnz}, emb 64, hops "
                                  "[64, 64, 64], mess_dropout 0.1 (device RNG), batch 1024, adam 0.05",
                      "last_loss": st.loss},
           "roofline": {"bound": "hbm", "algorithmic_bytes_per_step": bytes_step,
                        "achieved": bytes_step / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": bytes_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                        "gemm_tflops": flops / (dt / args.steps) / 1e12,
                        "note": "whole step (~35 launches); full-graph propagation per step like the reference"}}
    if not args.no_cpu_baseline:
        from oracle import torch_port

        w0 = {k: v.detach().cpu().numpy() for k, v in eng.model.state_dict().items()}
        host = [(users[k * Bn:(k + 1) * Bn].cpu(), pos[k * Bn:(k + 1) * Bn].cpu(), neg[k * Bn:(k + 1) * Bn].cpu())
                for k in range(8)]
        make = lambda: torch_port.TorchNGCFPort(w0, norm.coalesce(), [0.1] * L, 1e-5, Bn, "adam", 0.05)  # noqa: E731
        out["cpu_baseline"] = port_baseline(make, host, Bn, f"batch {Bn} on the same graph")
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--optimizer", default="adam", choices=["sgd", "adam", "rmsprop"],
                    help="mf: adam is the reference's own default (configs/mf_default.json); sgd / rmsprop are the other two torch_engine.py:23-39 builds")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--two-kernel", action="store_true",
                    help="mf: gradient kernel + dense optimizer sweep per step instead of the fused one-kernel step")
    ap.add_argument("--workload", default="mf", choices=["mf", "ncf", "lightgcn", "mf-c4shard", "mf-c4", "pgmf", "t2v", "ngcf"],
                    help="mf = BASELINE configs[1] (the headline); ncf = configs[2] (NeuMF, emb_dim 32)")
    ap.add_argument("--multi-gpu", default="auto", choices=["auto", "replicated", "sharded"],
                    help="N>1: replicate small tables (gradient all-reduce) or row-shard them "
                         "(all-to-all routing); auto = replicated below 64 MB of parameters")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    # HIPREC_BENCH_FORCE_SHARDED=1 exercises the N>1 code path on a single GPU (world size 1)
    dist_on = world > 1 or os.environ.get("HIPREC_BENCH_FORCE_SHARDED") == "1"
    if dist_on:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=device)

    import beta_recsys_amd as hp

    if args.workload == "ncf":
        return bench_ncf(args, device)
    if args.workload == "lightgcn":
        return bench_lightgcn(args, device)
    if args.workload in ("pgmf", "t2v"):
        return bench_siblings(args, device)
    if args.workload == "ngcf":
        return bench_ngcf(args, device)
    if args.workload in ("mf-c4shard", "mf-c4"):
        return bench_mf_c4shard(args, device, full=args.workload == "mf-c4")

    n_total = (args.warmup + args.steps) * B
    users, pos, neg = (t.to(device) for t in synth_triples(n_total, seed=100 + rank))
    torch.manual_seed(7 + rank)  # device-side randperm per epoch
    nw = args.warmup * B
    prepared = None
    if not dist_on:
        eng = make_engine(device, args.optimizer)
        eng.fused_step = not args.two_kernel
        warm = hp.DeviceTripleBatcher(users[:nw], pos[:nw], neg[:nw], B)
        timed = hp.DeviceTripleBatcher(users[nw:], pos[nw:], neg[nw:], B)
        if args.warmup > 0:
            eng.run_prepared_epoch(stage(eng, warm))
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        prepared = stage(eng, timed)  # per-epoch staging: permutation, per-batch sort by item, layout
        staging_s = time.perf_counter() - ts0
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        eng.run_prepared_epoch(prepared, sync=False)  # enqueues exactly args.steps steps
        ev1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        epoch_event_s = ev0.elapsed_time(ev1) * 1e-3
        assert len(timed) == args.steps
        st = eng.epoch_stats()
        assert st.step == args.warmup + args.steps, (st.step, args.warmup + args.steps)
        assert np.isfinite(st.loss_sum) and 0.3 < st.loss_sum / args.steps < 1.4, st.loss_sum
    else:
        # N > 1: tables row-sharded over the ranks (owner = row mod N), every rank feeds B triples
        # per step (weak scaling, global batch N*B), triples / item rows / item gradients are
        # routed with all-to-all over RCCL (beta-recsys_amd/sharded.py).
        from beta_recsys_amd.replicated import ReplicatedMFEngine
        from beta_recsys_amd.sharded import ShardedMFEngine

        cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str=str(device),
                             optimizer=args.optimizer, lr=LR, batch_size=B, loss="bpr"),
               "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
        param_bytes = 4 * ((U + I) * (D + 1) + 1)
        mode = args.multi_gpu
        if mode == "auto":
            mode = "replicated" if param_bytes < (64 << 20) else "sharded"
        torch.manual_seed(2020)
        with contextlib.redirect_stdout(io.StringIO()):
            seng = ShardedMFEngine(cfg) if mode == "sharded" else ReplicatedMFEngine(cfg)
        perm = torch.randperm(n_total, device=device)
        if mode == "replicated":  # stage the epoch like the single-GPU path: batches sorted by item
            from beta_recsys_amd.mf import sort_within_batches

            perm = sort_within_batches(perm, pos, B, I)
            seng.presorted = True
        users, pos, neg = users[perm], pos[perm], neg[perm]
        if mode == "sharded":
            step_fn = lambda batch: seng.train_single_batch(batch, sync=False)  # noqa: E731
            check_fn = seng.k.check_status
        else:
            # resident data-parallel epoch: one fused launch + one all-reduce per step; the sweep-only
            # flush that applies the last update is part of the timed region
            step_fn = None
            check_fn = seng.epoch_stats

        def run(lo, n_steps):
            last = None
            if mode != "sharded":
                pu, pp, pn = users.data_ptr(), pos.data_ptr(), neg.data_ptr()
                seng.fused_epoch_begin()
                for sidx in range(n_steps):
                    off = 8 * (lo + sidx * B)
                    seng.fused_step_ptr(pu + off, pp + off, pn + off, B)
                seng.fused_epoch_end()
                return last
            for sidx in range(n_steps):
                sl = slice(lo + sidx * B, lo + (sidx + 1) * B)
                last = step_fn((users[sl], pos[sl], neg[sl]))
            return last

        run(0, args.warmup)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(nw, args.steps)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = check_fn()
        if mode != "sharded":   # the replicas' loss sums are global: same sanity window as the single-GPU path
            assert st.step == args.warmup + args.steps, (st.step, args.warmup + args.steps)
            assert np.isfinite(st.loss_sum) and 0.3 < st.loss_sum / args.steps < 1.4, st.loss_sum
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if rank == 0:  # roofline of the dominant kernel: measured on a single-GPU engine
            eng = make_engine(device, args.optimizer)
            prepared = stage(eng, hp.DeviceTripleBatcher(users[:4 * B], pos[:4 * B], neg[:4 * B], B))
            if mode == "replicated":
                # the replicas run the same fused step kernel as the single-GPU epoch: its launch period
                # from HIP events around a short resident epoch on this rank's GPU
                n_ev = min(args.steps, 200)
                probe = stage(eng, hp.DeviceTripleBatcher(users[:n_ev * B], pos[:n_ev * B], neg[:n_ev * B], B))
                eng.run_prepared_epoch(probe, sync=False)
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                ev0.record()
                eng.run_prepared_epoch(probe, sync=False)
                ev1.record()
                torch.cuda.synchronize()
                epoch_event_s, event_steps = ev0.elapsed_time(ev1) * 1e-3, n_ev

    parallelism = "single GPU"
    if dist_on:
        parallelism = (f"row-sharded tables over {world} GPUs (owner = row mod {world}), all-to-all "
                       "routing of triples / item rows / item gradients over RCCL"
                       if mode == "sharded" else
                       f"dp{world}: replicated 2.5 MB tables, one fused launch + one RCCL all-reduce of the "
                       "dense gradient per step, global batch = N x 4096")
    if rank == 0:
        k_mean, k_med = kernel_timing(eng, prepared)
        bpt = algorithmic_bytes_per_triple(D)
        fused = eng.fused_step and (not dist_on or mode == "replicated")
        kind_id = {"sgd": 0, "adam": 1, "rmsprop": 2}[args.optimizer]
        if fused:
            # ONE kernel per step (gather + score + gradient scatter + the optimizer update of the
            # previous step): its launch period, from HIP events around the timed epoch, is what
            # the algorithmic bytes of a step are divided by
            dom_name = (f"mf_bpr_fused_kernel<1,{kind_id}> (gather + score + BPR grad + scatter + "
                        f"{args.optimizer} update, 1 launch/step)")
            dom_s = epoch_event_s / ((event_steps if dist_on else args.steps) + 1)
        else:
            dom_name = "mf_bpr_grad_kernel (gather + score + BPR grad + atomic scatter)"
            dom_s = k_mean
        achieved = bpt * B / dom_s / 1e9
        out = {
            "metric": "training interactions/sec (BPR triples)",
            "value": world * args.steps * B / dt,
            "unit": "triples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BPR-MF, MovieLens-1M-shaped synthetic (BASELINE configs[1]): 6040 users x "
                            "3706 items, dim 64, batch 4096 triples/GPU, uniform users, Zipf(1.0) "
                            "positives, uniform negatives",
                "optimizer": args.optimizer, "lr": LR, "loss": "bpr", "batch_per_gpu": B,
                "global_batch": B * world,
                "parallelism": parallelism,
            },
            "epoch_staging": None if dist_on else {
                "ms": staging_s * 1e3,
                "what": "per-epoch device-side batcher work outside the timed region: randperm, sort of "
                        "every batch by item, gather of the triples into visiting order",
                "value_including_staging": args.steps * B / (dt + staging_s),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom_name,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": bpt * B,
                "kernel_us": dom_s * 1e6,
                "grad_only_kernel_us": k_mean * 1e6,
                "traffic": measured_traffic_bytes(f"hiprec::mf_bpr_fused_kernel<1, {kind_id}, false>" if fused
                                                 else "hiprec::mf_bpr_grad_kernel<1>"),
                "step_frac": (world * args.steps * B / dt) / world * bpt / (HBM_PEAK_GBS * 1e9),
            },
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.optimizer)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
