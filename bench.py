"""bench.py — BPR-MF training throughput on MI355X (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--optimizer sgd|adam|rmsprop]
                    [--workload mf|mf-c4|mf-c4shard|ncf|lightgcn|...] [--scaling weak|strong]

A "step" is one pass of the hot path over one batch of B = 4096 synthetic (user, pos, neg) triples:
gather -> score -> BPR gradient -> scatter into the dense gradient -> optimizer update (exactly
the work of MFEngine.train_single_batch in the reference, beta_rec/models/mf.py:92-119).  The workload is
continuous training on the C2 data set of SURVEY.md §8(d): ~1 M triples per epoch (245 batches), resident in
HBM when the clock starts; the engine trains epoch after epoch of it exactly as TrainEngine._train drives
train_an_epoch.  Everything an epoch does on top of the resident triples is INSIDE the timed region: the
per-epoch device shuffle, the staging of the epoch (per-batch sort by item + layout; it runs on a side
stream while the previous epoch trains), the 245 fused steps and the flush launch that applies the last update.

Timing (>= 200 timed steps, median of >= 5 repeats): after W warm-up steps the next K steps
are timed, R = max(5, ceil(200 / K)) times in a row (each repeat continues where the last one stopped, across
epoch boundaries); `ms_per_step` is the MEDIAN over the repeats of (time of the K steps / K), `value` follows
from it, `steps` stays K and `repeats` says R; `wall_ms_per_step` is the plain wall clock of the whole R x K
region (launch latency of the first step included) for cross-checking.  N = 1: the repeats are enqueued back
to back and delimited by HIP events on the stream they run on (a host synchronize per 0.2 ms window would
measure the synchronize).  N > 1: the same events per rank, the whole R x K region bracketed by barrier +
torch.cuda.synchronize() on both sides, and the MAX over ranks taken per repeat afterwards.

`python bench.py --gpus N` without a torch.distributed environment starts its own N ranks (re-exec
through torch.distributed.run on 127.0.0.1); under torchrun it uses RANK / LOCAL_RANK / WORLD_SIZE as
given.  Rank 0 prints ONE JSON line.
"""
import argparse
import contextlib
import io
import json
import os
import socket
import statistics
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
MIN_TIMED_STEPS, MIN_REPEATS = 200, 5
EPOCH_STEPS = 245  # N = 1 000 000 triples per epoch -> 245 batches of 4096 (1 003 520 triples)
ROUND = "r06"


def algorithmic_bytes_per_triple(dim):
    """indices 3*8 B + 3 row reads + 3 row writes of (dim+1) fp32."""
    return 24 + 24 * (dim + 1)


def optimizer_sweep_bytes(optimizer, n_params):
    """what the dense optimizer adds per STEP on top of the per-triple row traffic --
    Adam reads w, m, v, g and writes w, m, v (28 P), RMSprop reads w, v, g and writes w, v (20 P); SGD's
    row updates are already the "3 row writes" of the per-triple figure."""
    return {"sgd": 0, "adam": 28, "rmsprop": 20}[optimizer] * n_params


def n_repeats(steps):
    return max(MIN_REPEATS, -(-MIN_TIMED_STEPS // max(steps, 1)))


def _barrier(group, device):
    """Barrier over the ranks of `group` through the engines' collective seam (torch.distributed, or the loopback
    world of the tests), bracketed by device synchronisation like the contract asks."""
    from beta_recsys_amd import _dist

    torch.cuda.synchronize()
    _dist.all_reduce(torch.zeros(1, dtype=torch.int32, device=device), group=group)
    torch.cuda.synchronize()


def timed_repeats(run_epoch, steps, device, dist_on=False, group=None):
    """Run `run_epoch(r)` (which enqueues exactly `steps` steps incl. their per-epoch staging) R times.
    Returns (per-repeat seconds, wall seconds of the whole region).  See the module docstring."""
    R = n_repeats(steps)
    if not dist_on:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(R + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs[0].record()
        for r in range(R):
            run_epoch(r)
            evs[r + 1].record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        return [evs[r].elapsed_time(evs[r + 1]) * 1e-3 for r in range(R)], wall
    from beta_recsys_amd import _dist

    # N > 1: the WHOLE region (R x K steps) is bracketed by barrier + synchronize on both sides; inside it every
    # rank delimits its K-step windows with HIP events on the stream the steps run on -- exactly like N = 1 -- and
    # the per-window times are MAX-reduced over the ranks afterwards.  (Bracketing every 20-step window with
    # synchronize + barrier + synchronize measured the brackets: 14.9 us/step at --steps 20 against 12.5 at 2000 for
    # the same world-1 replicated step)  The steps' own collectives keep the ranks in lock step.
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(R + 1)]
    _barrier(group, device)
    t0 = time.perf_counter()
    evs[0].record()
    for r in range(R):
        run_epoch(r)
        evs[r + 1].record()
    _barrier(group, device)
    wall = time.perf_counter() - t0
    per = [evs[r].elapsed_time(evs[r + 1]) * 1e-3 for r in range(R)]
    t = torch.tensor(per + [wall], dtype=torch.float64, device=device)
    _dist.all_reduce(t, op=_dist.ReduceOp.MAX, group=group)  # the slowest rank defines every repeat
    t = t.cpu().tolist()
    return t[:-1], t[-1]


def timing_fields(per, wall, steps, units_per_step, world=1):
    """The contract's fields from the per-repeat times: median epoch time / K."""
    med = statistics.median(per)
    return {"value": world * units_per_step * steps / med, "ms_per_step": med / steps * 1e3,
            "repeats": len(per), "timed_steps_total": len(per) * steps,
            "ms_per_step_by_repeat": [round(t / steps * 1e3, 5) for t in per],
            "ms_per_step_min": min(per) / steps * 1e3, "ms_per_step_max": max(per) / steps * 1e3,
            "wall_ms_per_step": wall / (len(per) * steps) * 1e3}


def self_launch(n):
    """`python bench.py --gpus N` without RANK / WORLD_SIZE: become `torch.distributed.run` with N ranks on
    this node (one process per GPU over RCCL), same arguments."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: needs {n} visible MI355X GPUs, this node shows {have} "
                         "(torch.cuda.device_count()); nothing was launched")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


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
    if os.environ.get("HIPREC_BENCH_SGD_MODE"):   # experiments: force MFEngine's sgd_mode (auto | rows | owned | dense)
        cfg["model"]["sgd_mode"] = os.environ["HIPREC_BENCH_SGD_MODE"]
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


def bench_ncf(args, device, world=1, rank=0, dist_on=False):
    """BASELINE configs[2]: NeuMF (GMF + MLP) on the ML-1M shape, batch 4096 (user, item, rating) samples with
    1 positive : 4 negatives, Adam lr 1e-3.  `--emb-dim 32` (default): tower 256->128->64->32 (MLP [128, 64, 32],
    quirk Q9); `--emb-dim 64`: tables 256/256/64/64, tower 512->256->128->64.  N > 1: replicated tables and tower,
    one all-reduce of [flat gradient | loss] per step (replicated.replicated_ncf_engine)."""
    import beta_recsys_amd as hp

    E, L, K, W = args.emb_dim, 3, args.steps, args.warmup
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=E, dropout=0.0, device_str=str(device),
                         optimizer="adam", lr=1e-3, batch_size=B, model="ncf_end",
                         mlp_config={"n_layers": L}, gmf_config={}),
           "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
    torch.manual_seed(2020)
    with contextlib.redirect_stdout(io.StringIO()):
        if dist_on and args.multi_gpu == "sharded":
            from beta_recsys_amd.sharded_ncf import ShardedNeuMFEngine

            eng = ShardedNeuMFEngine(cfg)   # the four tables row-sharded (owner = row mod N), tower replicated
            eng._enqueue_step = lambda u, i, r: eng.train_single_batch(u, i, r, sync=False)
            eng._sync_stats = lambda: type("S", (), {"loss": float(eng._g_ext[-2])})()
        elif dist_on:
            from beta_recsys_amd.replicated import replicated_ncf_engine

            eng = replicated_ncf_engine(hp.NeuMFEngine)(cfg)
        else:
            eng = hp.NeuMFEngine(cfg)
    n_total = (W + K) * B
    # the sample stream instance_bce_loader builds (data/base_data.py:182-216, num_negative 4): every
    # positive (Zipf item, rating 1) is followed by 4 negatives of the same user (uniform items, rating
    # 0), then the DataLoader shuffles the samples
    pu, ppos, pneg = synth_triples(n_total // 5 + 1, seed=100 + rank)
    g = torch.Generator().manual_seed(101 + rank)
    users = pu.repeat_interleave(5)[:n_total]
    items = torch.cat([ppos[:, None], torch.randint(0, I, (ppos.numel(), 4), generator=g)], 1).reshape(-1)[:n_total]
    ratings = torch.tensor([1.0, 0, 0, 0, 0]).repeat(ppos.numel())[:n_total]
    shuffle = torch.randperm(n_total, generator=g)
    users, items, ratings = (t[shuffle].contiguous().to(device) for t in (users, items, ratings))

    def run(lo, n):
        for k in range(n):
            sl = slice(lo + k * B, lo + (k + 1) * B)
            eng._enqueue_step(users[sl], items[sl], ratings[sl])

    run(0, W)
    per, wall = timed_repeats(lambda r: run(W * B, K), K, device, dist_on)
    st = eng._sync_stats()
    if rank != 0:
        return None
    traffic, traffic_src = traffic_step_from_profiles("ncf" if E == 32 else "ncf64")
    dims = [2 * E * 2 ** (L - 1)] + [E * 2 ** (L - 1 - i) for i in range(L)]      # 256,128,64,32 at E 32
    macs = sum(a * b for a, b in zip(dims[:-1], dims[1:])) + 2 * E                # tower + head (E mlp + E mf)
    flops = 3 * 2 * macs                                                          # fwd + dgrad + wgrad per sample
    out = {"metric": "training interactions/sec (NCF samples)", "unit": "samples/s"}
    out.update(timing_fields(per, wall, K, B, world))
    step_s = out["ms_per_step"] * 1e-3
    out.update({"n_gpus": world, "steps": K, "warmup": W, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"NeuMF (BASELINE configs[2]): 6040 x 3706, emb_dim {E} => tables "
                                       f"{dims[0] // 2}/{dims[0] // 2}/{E}/{E}, tower {'->'.join(map(str, dims))}, head "
                                       f"{2 * E}->1, batch 4096/GPU, adam 1e-3",
                           "parallelism": (f"tables row-sharded over {world} GPUs (owner = row mod {world}), ids / rows / row "
                                           "gradients routed with RCCL all-to-alls, tower replicated + all-reduced"
                                           if dist_on and args.multi_gpu == "sharded" else
                                           f"dp{world}: replicated tables + tower, one RCCL all-reduce per step" if dist_on
                                           else "single GPU"), "rccl_world_size": world if dist_on else None,
                           "last_loss": st.loss},
                "roofline": {"bound": "mfma", "achieved": B * flops / step_s / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                             "frac": B * flops / step_s / 1e12 / 157.3, "flops_per_sample": flops,
                             "note": "whole step (all launches) against the dense fp32 MFMA peak; traffic = HBM bytes of "
                                     "the step's launches from the committed PMC passes",
                             "traffic": traffic, "traffic_source": traffic_src}})
    out["roofline"].update(dominant_kernel_from_profiles("ncf" if E == 32 else "ncf64"))
    if not args.no_cpu_baseline and world == 1:
        from oracle.torch_port import TorchNeuMFPort   # the reference's ATen op sequence (ncf.py:52-71, 100-120)

        w0 = {k: v.detach().cpu().numpy() for k, v in eng.model.state_dict().items()}
        cu, ci, cr = (t[: 16 * B].cpu() for t in (users, items, ratings))
        batches = [(cu[k * B:(k + 1) * B], ci[k * B:(k + 1) * B], cr[k * B:(k + 1) * B]) for k in range(16)]
        out["cpu_baseline"] = port_baseline(lambda: TorchNeuMFPort(w0, "adam", 1e-3), batches, B,
                                            f"NeuMF emb_dim {E}, batch {B}", unit="samples/s")
    return out


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
    c4opt = args.c4_optimizer                 # sgd (primary) | adam | rmsprop (dense-Adam secondary)
    owned = args.sgd_mode in ("owned", "owned_atomic") and c4opt == "sgd"
    cfg = {"model": dict(n_users=Uc, n_items=Ic, emb_dim=Dc, device_str=str(device), optimizer=c4opt,
                         lr=LR, batch_size=Bc, loss="bpr", sgd_mode=args.sgd_mode, dense_opt=args.dense_opt,
                         lazy_grad=args.lazy_grad),
           "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
    torch.manual_seed(2020)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = hp.MFEngine(cfg)
    steps, warm = min(args.steps, 100), min(args.warmup, 10)
    full_cov = args.epoch_coverage == "full"
    # "sample": 50 steps = 3.3 M triples per epoch, users drawn uniformly -- at the whole-table size an epoch meets
    # 28 % of the users.  "full": EVERY user and item row occurs in every epoch (a data set in which every user has
    # interactions, as the reference's do): the epoch is as long as that takes (whole table: 153 steps = 10 M triples),
    # and what the exact lazy optimizers replay per step is the whole table's worth of zero-gradient steps.
    epoch_steps = max(50, -(-max(Uc, Ic) // Bc)) if full_cov else 50
    g = torch.Generator().manual_seed(5)
    n_total = epoch_steps * Bc
    users = torch.randint(0, Uc, (n_total,), generator=g)
    pz = 1.0 / torch.arange(1, Ic + 1, dtype=torch.float64)
    pos = torch.randperm(Ic, generator=g)[torch.multinomial(pz / pz.sum(), n_total, True, generator=g)]
    if full_cov:     # one occurrence of every row, anywhere in the epoch (the batcher reshuffles every epoch anyway)
        users[torch.randperm(n_total, generator=g)[:Uc]] = torch.randperm(Uc, generator=g)
        pos[torch.randperm(n_total, generator=g)[:Ic]] = torch.randperm(Ic, generator=g)
    neg = torch.randint(0, Ic, (n_total,), generator=g)

    def met(n_rows, *ids):     # fraction of a table's rows an epoch meets (host arithmetic: nothing of it on the GPU)
        seen = torch.zeros(n_rows, dtype=torch.bool)
        for t in ids:
            seen[t] = True
        return float(seen.sum()) / n_rows

    coverage = {"users": met(Uc, users), "items": met(Ic, pos, neg)}
    users, pos, neg = users.to(device), pos.to(device), neg.to(device)
    loader = hp.DeviceTripleBatcher(users, pos, neg, Bc)
    torch.manual_seed(7)
    # continuous training, epoch after epoch, K-step windows (see bench_mf): the staging of an epoch -- device
    # shuffle, per-batch sort by item, layout and (owned-rows step) the row-ownership arrays, all torch sorts at
    # this batch size -- runs on the side stream while the previous epoch trains
    state = {"pos": 0, "prepared": None}

    def advance(n):
        while n > 0:
            take = min(n, epoch_steps - state["pos"])
            if state["pos"] == 0:
                state["prepared"] = eng.prepare_epoch(loader)
            eng.run_prepared_epoch(state["prepared"], sync=False, prefetch=loader,
                                   steps=(state["pos"], state["pos"] + take))
            state["pos"] = (state["pos"] + take) % epoch_steps
            n -= take

    advance(warm)
    per, wall = timed_repeats(lambda r: advance(steps), steps, device)
    advance((epoch_steps - state["pos"]) % epoch_steps)
    # two WHOLE epochs of continuous training between one event pair (staging on the side stream, every step, the
    # epoch's flush): what a K-step window's median hides when K is shorter than an epoch
    w0_, w1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0_.record()
    advance(2 * epoch_steps)
    w1_.record()
    torch.cuda.synchronize()
    whole_epochs_s = w0_.elapsed_time(w1_) * 1e-3 / (2 * epoch_steps)
    st = eng.epoch_stats()
    # the same steps with the epoch staged beforehand (nothing else on the GPU): the step kernels alone
    prepared = eng.prepare_epoch(loader)
    eng._drop_prefetch()
    torch.cuda.synchronize()
    eng.run_prepared_epoch(prepared, sync=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.run_prepared_epoch(prepared, sync=False)
    e1.record()
    torch.cuda.synchronize()
    alone_s = e0.elapsed_time(e1) * 1e-3 / epoch_steps
    eng.epoch_stats()
    bpt = algorithmic_bytes_per_triple(Dc)
    if owned:
        # owner pulls (round 5): the gradient launch + the apply launch of a step, timed together -- the bytes of
        # SURVEY 8d are the step's, so the period they are divided by is the step's two launches
        kname = ("mf_bpr_owned_kernel<2> (1 launch/step, shared rows through device-scope atomics)"
                 if args.sgd_mode == "owned_atomic" else
                 "mf_bpr_owned_kernel<2,false,false,true> (gather + score + BPR grad; rows with one contributor "
                 "updated in place, the others' parts stored) + pull_apply_vec_kernel<32> (one lane group per shared "
                 "row sums its parts and stores w - lr g): 2 launches/step, no float atomics")
        k_s = alone_s
        traffic, traffic_src = traffic_step_from_profiles("mf-c4" if full else "mf-c4shard", "pull_apply")
    elif getattr(eng, "_lazy", None) is not None:
        kname, k_s, traffic, traffic_src = None, alone_s, None, None   # the lazy step's three launches: named below
    else:
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
        kname, k_s = "mf_bpr_grad_kernel<2>", a.elapsed_time(b) / 50 * 1e-3
        traffic, traffic_src = (None, None) if full else traffic_from_profiles("hiprec::mf_bpr_grad_kernel<2>", "mf-c4shard")
        if c4opt != "sgd":      # the timed calls advanced the optimizer clock and left a gradient behind
            eng._g_flat.zero_()
            eng._lazy_mark_current()
    lazy = getattr(eng, "_lazy", None) is not None
    if c4opt != "sgd":
        # SURVEY 8d prices the reference's dense optimizer at 28 P (Adam) / 20 P (RMSprop) bytes per step; the exact lazy
        # form moves the step's rows instead: w, m, v of its 3 rows read and written (RMSprop: w, v) on top of the
        # gradient round trip the SGD figure counts.  The roofline below is on the bytes the form that ran moves.
        sweep_bytes = optimizer_sweep_bytes(c4opt, eng.model.flat.numel())
        row_bytes = {"adam": 3 * 6, "rmsprop": 3 * 4}[c4opt] * 4 * (Dc + 1)
        bpt_run = bpt + (row_bytes if lazy else sweep_bytes / Bc)
        kname = (("lazy step: catch-up + mf_bpr_owned_kernel<2,false,false,true> (gradient parts -> contribution buffer) + "
                  "lazy_pull_apply_kernel (sum, replay the moments, step, stamp): 3 launches, no dense gradient traffic"
                  if eng._lazy_owned() == "pull" else
                  "lazy step: catch-up + mf_bpr_owned_kernel<2,false,true> (gradients) + update (3 launches)") if lazy
                 else "mf_bpr_fused_kernel / dense sweep")
        k_s = alone_s
        traffic, traffic_src = (traffic_step_from_profiles(("mf-c4_" if full else "mf-c4shard_") + c4opt,
                                                              ("mf_bpr_owned_kernel", "mf_bpr_grad_kernel"))
                                if lazy else (None, None))   # the committed PMC passes ran the lazy form
    else:
        sweep_bytes, bpt_run = 0, bpt
    out = {"metric": "training interactions/sec (BPR triples)", "unit": "triples/s"}
    out.update(timing_fields(per, wall, steps, Bc))
    out.update({"n_gpus": 1, "steps": steps, "warmup": warm,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": ("BPR-MF, BASELINE configs[3] whole on one GPU: 10M x 1M rows, dim 128, batch "
                                        "65536, " if full else
                                        "BPR-MF, one rank's shard of BASELINE configs[3]: 1.25M x 125k rows, dim 128, "
                                        "batch 65536 (no exchange timed), ") +
                                       ("plain SGD" if c4opt == "sgd" else
                                        f"{c4opt}, " + ("exact lazy replay (csrc/lazy_opt.hip): catch-up + gradient kernel + "
                                                        "update of the batch's rows per step, one flush per epoch"
                                                        if eng._lazy is not None else "dense sweep every step")),
                           "optimizer": c4opt,
                           "sgd_mode": args.sgd_mode,
                           "epoch": f"{epoch_steps} steps = {n_total} triples",
                           "epoch_coverage": args.epoch_coverage,
                           "rows_met_per_epoch": coverage,
                           "timed_region": "continuous training; per epoch one staging pass (device shuffle, per-batch "
                                           "sort, layout, row ownership) on a side stream during the previous epoch",
                           "ms_per_step_kernels_alone": alone_s * 1e3,
                           "ms_per_step_whole_epochs": whole_epochs_s * 1e3,
                           "last_loss": st.loss},
                "roofline": {"bound": "hbm", "kernel": kname, "achieved": bpt_run * Bc / k_s / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bpt_run * Bc / k_s / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": bpt_run * Bc, "kernel_us": k_s * 1e6,
                             "dense_sweep_bytes_per_launch": bpt * Bc + sweep_bytes,
                             # SURVEY 8d's own denominator for the optimizer that ran (28 P / 20 P of a dense sweep on
                             # top of the rows): the rate a dense sweep would have to run at to match this step
                             "dense_equivalent_GBps": (bpt * Bc + sweep_bytes) / k_s / 1e9,
                             "traffic": traffic, "traffic_source": traffic_src,
                             "step_frac": out["value"] * bpt_run / (HBM_PEAK_GBS * 1e9)}})
    if not args.no_cpu_baseline:
        # the reference's CPU path at this size: nn.Embedding is non-sparse, so every step materialises DENSE gradients
        # of both tables (mf.py:117) and torch.optim.SGD sweeps them -- a bounded sample of a few steps
        from oracle.torch_port import TorchMFPort

        del eng, prepared
        torch.cuda.empty_cache()
        gcpu = torch.Generator().manual_seed(3)
        w0 = {"user_emb.weight": torch.empty(Uc, Dc).normal_(0, 0.1, generator=gcpu),
              "item_emb.weight": torch.empty(Ic, Dc).normal_(0, 0.1, generator=gcpu),
              "user_bias.weight": torch.zeros(Uc, 1), "item_bias.weight": torch.zeros(Ic, 1),
              "global_bias": torch.zeros(1)}
        cu, cp, cn = (t[: 4 * Bc].cpu() for t in (users, pos, neg))
        batches = [(cu[k * Bc:(k + 1) * Bc], cp[k * Bc:(k + 1) * Bc], cn[k * Bc:(k + 1) * Bc]) for k in range(4)]
        nt = min(32, torch.get_num_threads())
        out["cpu_baseline"] = port_baseline(lambda: TorchMFPort(w0, c4opt, LR, "bpr"), batches, Bc,
                                            f"BPR-MF {Uc} x {Ic} x {Dc}, batch {Bc}, torch.optim ({c4opt}) over dense gradients",
                                            budget_s=10.0, thread_counts=[nt])
    return out


def bench_mf_c4_sharded(args, device, world, rank, group=None):
    """BASELINE configs[3] as specified: 10M users x 1M items, dim 128, tables ROW-SHARDED over the ranks
    (owner = row mod N), every rank feeds 65536 triples per step (global batch N x 65536), triples / item rows /
    item-row gradients routed with RCCL all-to-alls, exact SGD on the rows each shard's step touched."""
    from beta_recsys_amd.sharded import ShardedMFEngine

    Uc, Ic, Dc, Bc = 10_000_000, 1_000_000, 128, 65536
    cfg = {"model": dict(n_users=Uc, n_items=Ic, emb_dim=Dc, device_str=str(device), optimizer=args.c4_optimizer, lr=LR,
                         batch_size=Bc, loss="bpr", sgd_mode="rows", shard_init="local", step_driver=args.step_driver,
                         dense_opt=args.dense_opt, shard_sgd=args.shard_sgd,
                         shard_self_exchange=bool(getattr(args, "force_exchange", False))),
           "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
    torch.manual_seed(2020)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ShardedMFEngine(cfg, process_group=group)
    eng._step_comm()   # (collective) the step driver's communicator is created here, not inside a timed window
    if not args.no_plan_prefetch:
        eng.prefetch_setup()   # likewise the side stream / process group of the prefetched plans
    import beta_recsys_amd as hp

    steps, warm = min(args.steps, 100), min(args.warmup, 10)
    epoch_steps = 50
    n_total = epoch_steps * Bc
    g = torch.Generator().manual_seed(5 + rank)
    users = torch.randint(0, Uc, (n_total,), generator=g).to(device)
    pz = 1.0 / torch.arange(1, Ic + 1, dtype=torch.float64)
    item_perm = torch.randperm(Ic, generator=torch.Generator().manual_seed(5))   # the same popular items on every rank
    pos = item_perm[torch.multinomial(pz / pz.sum(), n_total, True, generator=g)].to(device)
    neg = torch.randint(0, Ic, (n_total,), generator=g).to(device)
    loader = hp.DeviceTripleBatcher(users, pos, neg, Bc)
    torch.manual_seed(7 + rank)
    state = {"pos": 0, "plan": None}

    def advance(n):   # continuous training; an epoch is routed once (plan_epoch), then run step by step
        while n > 0:
            take = min(n, epoch_steps - state["pos"])
            if state["pos"] == 0:
                state["plan"] = eng.take_plan(loader)   # prefetched during the previous epoch, if there was one
            eng.run_planned_epoch(state["plan"], steps=(state["pos"], state["pos"] + take), sync=False)
            state["pos"] = (state["pos"] + take) % epoch_steps
            if state["pos"] == 0 and not args.no_plan_prefetch:
                eng.prefetch_plan(loader)   # the whole epoch is enqueued: route the next one on the side stream
            n -= take

    advance(warm)
    per, wall = timed_repeats(lambda r: advance(steps), steps, device, dist_on=True, group=group)
    eng.k.check_status()
    exchange = plan_exchange_bytes(state["plan"], Dc)
    if rank != 0:
        return None
    bpt = algorithmic_bytes_per_triple(Dc)
    lazy = eng._lazy is not None
    opt = args.c4_optimizer
    # the reference's dense optimizer adds 28 P (Adam) / 20 P (RMSprop) bytes per step on this rank's P
    # parameters.  The exact lazy form moves the step's rows instead: per touched row w, m, v read and written
    # (RMSprop: w, v) on top of the gradient round trip the SGD figure already counts.
    p_local = eng.model.flat.numel()
    sweep_bytes = optimizer_sweep_bytes(opt, p_local)
    row_bytes = {"sgd": 0, "adam": 3 * 6, "rmsprop": 3 * 4}[opt] * 4 * (Dc + 1)   # per triple: 3 rows
    moved_bpt = bpt + (row_bytes if lazy else sweep_bytes / Bc)
    out = {"metric": "training interactions/sec (BPR triples)", "unit": "triples/s"}
    out.update(timing_fields(per, wall, steps, Bc, world))
    out.update({"n_gpus": world, "steps": steps, "warmup": warm, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "BPR-MF, BASELINE configs[3]: 10M x 1M rows, dim 128, batch 65536 triples/GPU, "
                                       "uniform users, Zipf(1.0) positives, " +
                                       ("exact SGD on touched rows" + (", owner pulls (no float atomics)"
                                                                        if eng._pull_steps() else "") if opt == "sgd" else
                                        f"exact lazy {opt}: the step's rows are caught up / stepped, lagging rows replayed "
                                        "bit-identically to the dense sweep, one flush per 50-step epoch inside the clock"
                                        if lazy else f"dense {opt} sweep of every shard per step"),
                           "parallelism": f"tables row-sharded over {world} GPUs (owner = row mod {world}); the epoch is "
                                          "routed once by the planner kernels (triples -> owner(user), de-duplicated item "
                                          "requests -> owner(item)); per step 2 exact-size exchanges (rows out, gradients + "
                                          "loss partials back) around 4 launches, " +
                                          ("enqueued from C with grouped ncclSend/ncclRecv" if eng._step_mode == "c"
                                           else "through torch.distributed.all_to_all_single"),
                           "step_driver": eng._step_mode, "optimizer": args.c4_optimizer,
                           "global_batch": world * Bc, "rccl_world_size": world,
                           "exchange_bytes_per_step": round(exchange["exchange_bytes_per_step"]),
                           "exchange_bytes_per_step_off_gpu": round(exchange["exchange_bytes_per_step_off_gpu"]),
                           "a2a_GBps_per_gpu": exchange["exchange_bytes_per_step_off_gpu"] / (out["ms_per_step"] * 1e-3) / 1e9,
                           **self_exchange_fields(eng, exchange, out["ms_per_step"] * 1e-3)},
                "roofline": {"bound": "hbm", "kernel": "whole sharded step (per GPU)",
                             "achieved": out["value"] / world * moved_bpt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": out["value"] / world * moved_bpt / (HBM_PEAK_GBS * 1e9),
                             "algorithmic_bytes_per_launch": moved_bpt * Bc,
                             "bytes_model": ("rows of the step: 24 + 24 (D+1) per triple" +
                                             ("" if opt == "sgd" else
                                              f" + {row_bytes} (w, m, v of its 3 rows read and written)" if lazy else
                                              f" + the dense sweep's {sweep_bytes} per step")),
                             # what SURVEY 8d prices the reference's dense optimizer at, for comparison
                             "dense_sweep_bytes_per_launch": bpt * Bc + sweep_bytes, "traffic": None}})
    return out


def c5_graph():
    """ML-1M degree profile, 988 k UNIQUE train edges => nnz(D^-1 (A + I)) = 2 x 988 000 + 9 746 =
    1.99 M (rounds 1-2 drew 1 M edges with duplicates: 745 k unique, nnz 1.49 M).  Zipf items, duplicates redrawn.
    The reference's norm_adj = D^-1 (A + I) over users + items (data/deprecated_data_base.py:331-353 +
    utils/common_util.py normalized_adj_single), built once on the host like the reference does; scipy COO."""
    import scipy.sparse as sp

    rng = np.random.default_rng(0)
    n_edges = 988_000
    p = 1.0 / np.arange(1, I + 1) ** 0.9
    p /= p.sum()
    item_of = rng.permutation(I)
    pairs = np.zeros(0, dtype=np.int64)
    while pairs.size < n_edges:
        m = int((n_edges - pairs.size) * 1.3) + 1000
        pairs = np.unique(np.concatenate([pairs, rng.integers(0, U, m) * I + item_of[rng.choice(I, m, p=p)]]))
    pairs = rng.permutation(pairs)[:n_edges]
    eu, ei = pairs // I, pairs % I
    n_nodes = U + I
    rows, cols = np.concatenate([eu, ei + U]), np.concatenate([ei + U, eu])
    a = sp.coo_matrix((np.ones(len(rows), dtype=np.float32), (rows, cols)), shape=(n_nodes, n_nodes)).tocsr()
    a = a + sp.eye(n_nodes, dtype=np.float32, format="csr")
    adj = sp.diags(1.0 / np.asarray(a.sum(1)).flatten()).dot(a).astype(np.float32).tocoo()
    assert adj.nnz == 2 * n_edges + n_nodes
    return adj


def bench_lightgcn(args, device, world=1, rank=0, dist_on=False):
    """BASELINE configs[4]: LightGCN on an ML-1M-sized graph (~1M interactions, nnz ~2M), 3 layers,
    dim 64, batch 1024 triples, keep_pro 0.6 (device-side edge dropout), Adam lr 0.05."""
    import beta_recsys_amd as hp

    L, Bl = 3, 1024
    adj = c5_graph()
    idx = torch.from_numpy(np.vstack((adj.row, adj.col)).astype(np.int64))
    norm = torch.sparse_coo_tensor(idx, torch.from_numpy(adj.data), torch.Size(adj.shape))
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, layer_size=[D] * L, keep_pro=0.6, regs=[1e-5],
                         device_str=str(device), optimizer="adam", lr=0.05, batch_size=Bl, norm_adj=norm,
                         dropout_rng="device"),
           "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
    if getattr(args, "lane_slots", None):
        cfg["model"]["spmm_lane_slots"] = args.lane_slots
    torch.manual_seed(2020)
    with contextlib.redirect_stdout(io.StringIO()):
        if dist_on:   # data-parallel replicas: every rank its own 1024 triples of the global batch, one all-reduce
            from beta_recsys_amd.replicated import replicated_flat_engine

            cfg["model"]["dropout_seed"] = 11       # every replica drops the same edges in a step
            cfg["model"]["dp_collective"] = args.dp_collective
            eng = replicated_flat_engine(hp.LightGCNEngine)(cfg)
            eng._setup()                            # (collective) communicator created outside the timed windows
        else:
            eng = hp.LightGCNEngine(cfg)
    K, W = args.steps, args.warmup
    n_total = (W + K) * Bl
    users, pos, neg = (t.to(device) for t in synth_triples(n_total, seed=100 + rank))

    def run(lo, n):
        for k in range(n):
            sl = slice(lo + k * Bl, lo + (k + 1) * Bl)
            eng._enqueue_step((users[sl], pos[sl], neg[sl]))

    run(0, W)
    per, wall = timed_repeats(lambda r: run(W * Bl, K), K, device, dist_on)
    st = eng._sync_stats()
    if rank != 0:
        return None
    nnz, N = adj.nnz, U + I
    traffic, traffic_src = traffic_step_from_profiles("lightgcn", "lightgcn_loss_kernel")   # one loss launch per step
    # 2L SpMMs x [nnz*(4+4) + (N+1)*8 + 2*N*D*4] bytes (+ the keep byte per edge)
    bytes_step = 2 * L * (nnz * 9 + (N + 1) * 8 + 2 * N * D * 4)
    out = {"metric": "training interactions/sec (LightGCN triples)", "unit": "triples/s"}
    out.update(timing_fields(per, wall, K, Bl, world))
    step_s = out["ms_per_step"] * 1e-3
    out.update({"n_gpus": world, "steps": K, "warmup": W, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"LightGCN (BASELINE configs[4]): 6040 x 3706 graph, nnz {nnz}, 3 layers, "
                                       "dim 64, batch 1024, keep_pro 0.6 (device RNG), adam 0.05",
                           # full-graph propagation per step => replicas only, no data-path collective
                           "parallelism": (f"dp{world}: the full graph replicated (SURVEY 8e: replicas only), every rank "
                                           f"{Bl} triples of the global batch, one RCCL all-reduce of the 2.5 MB gradient "
                                           "per step" if dist_on else "single GPU"),
                           "global_batch": world * Bl,
                           "last_loss": st.loss},
                "roofline": {"bound": "hbm", "algorithmic_bytes_per_step": bytes_step,
                             "achieved": bytes_step / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": bytes_step / step_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                             "traffic_source": traffic_src}})
    out["roofline"].update(dominant_kernel_from_profiles("lightgcn"))
    if not args.no_cpu_baseline and world == 1:
        from oracle.torch_port import TorchLightGCNPort   # torch.sparse.mm, lightgcn.py:46-78, 119-152

        w0 = {k: v.detach().cpu().numpy() for k, v in eng.model.state_dict().items()}
        cu, cp, cn = (t[: 8 * Bl].cpu() for t in (users, pos, neg))
        batches = [(cu[k * Bl:(k + 1) * Bl], cp[k * Bl:(k + 1) * Bl], cn[k * Bl:(k + 1) * Bl]) for k in range(8)]
        out["cpu_baseline"] = port_baseline(lambda: TorchLightGCNPort(w0, norm, L, 0.6, 1e-5, "adam", 0.05), batches, Bl,
                                            f"LightGCN on the same graph (nnz {nnz}), batch {Bl}")
    return out


def port_baseline(make_port, batches, units_per_step, what, budget_s=9.0, unit="triples/s", thread_counts=None):
    """cpu_baseline for the sibling workloads: the reference's ATen op sequence (oracle/torch_port.py,
    pinned on goldens from the real reference) on this box's host cores, bounded sample, best of a few
    intra-op thread counts (cores = the count that won)."""
    all_threads = torch.get_num_threads()
    candidates = thread_counts or sorted({all_threads, min(32, all_threads), min(8, all_threads)}, reverse=True)
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
    return {"value": steps * units / dt, "unit": unit, "cores": nt, "kind": "port",
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
    per, wall = timed_repeats(lambda r: run(args.warmup * Bs, args.steps), args.steps, device)
    dt = statistics.median(per)   # median over the repeats of one {args.steps}-step pass
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
    out.update(repeats=len(per), timed_steps_total=len(per) * args.steps,
               wall_ms_per_step=wall / (len(per) * args.steps) * 1e3)
    return out


def bench_ngcf(args, device):
    """NGCF at configs/ngcf_default.json (emb 64, three hops of 64, mess_dropout 0.1,
    batch 1024, Adam lr 0.05) on the same ML-1M-sized graph as the LightGCN workload."""
    import beta_recsys_amd as hp

    L, Bn = 3, 1024
    adj = c5_graph()      # 988 k unique edges, nnz 1.99 M (rounds 1-2: 1.49 M)
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
    per, wall = timed_repeats(lambda r: run(args.warmup * Bn, args.steps), args.steps, device)
    dt = statistics.median(per)   # median over the repeats of one {args.steps}-step pass
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
    out["roofline"]["traffic"], out["roofline"]["traffic_source"] = traffic_step_from_profiles("ngcf")
    out["roofline"].update(dominant_kernel_from_profiles("ngcf"))
    if not args.no_cpu_baseline:
        from oracle import torch_port

        w0 = {k: v.detach().cpu().numpy() for k, v in eng.model.state_dict().items()}
        host = [(users[k * Bn:(k + 1) * Bn].cpu(), pos[k * Bn:(k + 1) * Bn].cpu(), neg[k * Bn:(k + 1) * Bn].cpu())
                for k in range(8)]
        make = lambda: torch_port.TorchNGCFPort(w0, norm.coalesce(), [0.1] * L, 1e-5, Bn, "adam", 0.05)  # noqa: E731
        out["cpu_baseline"] = port_baseline(make, host, Bn, f"batch {Bn} on the same graph")
    out.update(repeats=len(per), timed_steps_total=len(per) * args.steps,
               wall_ms_per_step=wall / (len(per) * args.steps) * 1e3)
    return out


_EVIDENCE = {}


def evidence_stamp(workload=None):
    """{"commit", "source_hash", "stale", "group"} for the committed profiles of `workload` (a profile file suffix:
    adam, mf-c4shard_adam, ncf64 ...).  tools/collect_profiles.py leaves profiles/rNN_stamp.json: per evidence group
    (__graft_entry__.EVIDENCE_GROUPS: the source files a workload's kernels are built from) the commit the group
    was measured at and the sha256 of each of those files as measured.  A number read from profiles/ is attached to
    a bench line only when (i) the library THIS run loaded was built from the sources of this tree
    (`hiprec_source_hash()` == the hash of csrc/ + include/) and (ii) every file of the workload's group is
    byte-identical to what was measured; otherwise the line says `stale` and carries no profile-sourced number
    (a kernel changed after the last refresh used to ride along silently with last round's counters)."""
    import __graft_entry__ as entry

    group = entry.evidence_group(workload)
    if group not in _EVIDENCE:
        st = {"commit": None, "source_hash": None, "stale": True, "group": group}
        try:
            with open(os.path.join(ROOT, "profiles", f"{ROUND}_stamp.json")) as f:
                rec = json.load(f)["groups"][group]
            st["commit"], st["source_hash"] = rec.get("commit"), rec.get("source_hash")
            from beta_recsys_amd import _lib

            if "lib_matches_tree" not in _EVIDENCE:
                _EVIDENCE["lib_matches_tree"] = _lib.load().hiprec_source_hash().decode() == entry.source_hash()
                _EVIDENCE["tree_files"] = entry.source_file_hashes()
            now = _EVIDENCE["tree_files"]
            st["stale"] = not (_EVIDENCE["lib_matches_tree"] and
                               all(now.get(f) == rec["files"].get(f) for f in entry.EVIDENCE_GROUPS[group]))
        except Exception:
            pass
        _EVIDENCE[group] = st
    return _EVIDENCE[group]


def stamp_roofline(out, workload=None):
    """Name, in the line itself, which build the profile-sourced fields of `roofline` belong to."""
    roof = out.get("roofline") if out else None
    if roof is None:
        return out
    st = evidence_stamp(workload)
    from_profiles = any(str(roof.get(k) or "").startswith("profiles/") for k in ("traffic_source", "kernel_us_source"))
    roof["traffic_commit"] = st["commit"] if from_profiles else None
    roof["evidence_group"] = st["group"]
    roof["stale"] = bool(st["stale"])
    return out


def traffic_from_profiles(kernel, workload=None):
    """HBM bytes per launch (FETCH_SIZE + WRITE_SIZE) of a kernel from the COMMITTED rocprofv3 PMC passes
    of THIS round (none when they were taken with another build of the sources: evidence_stamp) as (bytes | None,
    source file | None).  It is not measured in this run: counters need their own rocprofv3 passes (tools/pmc_workload.sh)."""
    if evidence_stamp(workload)["stale"]:
        return None, None
    for rnd in (ROUND,):
        try:
            if workload is None:
                name = f"{rnd}_pmc_summary.json"
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    k = json.load(f)[kernel]
                return (k["FETCH_SIZE_KB_mean"] + k["WRITE_SIZE_KB_mean"]) * 1024.0, "profiles/" + name
            name = f"{rnd}_pmc_other_workloads.json"
            with open(os.path.join(ROOT, "profiles", name)) as f:
                k = json.load(f)[workload][kernel]
            return (k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0, "profiles/" + name
        except Exception:
            continue
    return None, None


def dominant_kernel_from_profiles(workload):
    """The kernel a workload spends most of its GPU time in, with its average launch duration, from the COMMITTED
    `rocprofv3 --kernel-trace --stats` summary of the same bench command (profiles/rNN_kernel_stats_<workload>.csv:
    rows sorted by total duration): {"kernel", "kernel_us", "kernel_share_pct", "kernel_us_source"} (empty when no
    summary is there yet).  Not measured in this run -- the step is timed live, its kernels by the profiler."""
    import csv

    if evidence_stamp(workload)["stale"]:
        return {}
    for rnd in (ROUND,):
        name = f"{rnd}_kernel_stats_{workload}.csv"
        try:
            with open(os.path.join(ROOT, "profiles", name), newline="") as f:
                rows = [r for r in csv.DictReader(f) if "hiprec::" in r["Name"]]
            top = rows[0]
            return {"kernel": top["Name"].split("(")[0].replace("void ", "").strip(),
                    "kernel_us": float(top["AverageNs"]) / 1e3, "kernel_calls": int(top["Calls"]),
                    "kernel_share_pct": float(top["Percentage"]), "kernel_us_source": "profiles/" + name}
        except Exception:
            continue
    return {}


def traffic_step_from_profiles(workload, step_kernel="opt_dense_kernel"):
    """HBM bytes per STEP of a workload: (FETCH_SIZE + WRITE_SIZE) x launches per step summed over its kernels, from
    the committed per-workload PMC passes (profiles/rNN_pmc_other_workloads.json: KB per dispatch and the number
    of dispatches of every kernel).  (bytes | None, source | None).  step_kernel: a kernel that runs exactly once per
    step (its dispatch count is the number of steps)."""
    if evidence_stamp(workload)["stale"]:
        return None, None
    for rnd in (ROUND,):
        try:
            name = f"{rnd}_pmc_other_workloads.json"
            with open(os.path.join(ROOT, "profiles", name)) as f:
                ks = json.load(f)[workload]
            # every workload ends its step with ONE dense optimizer sweep: its dispatch count is the number of steps
            names = (step_kernel,) if isinstance(step_kernel, str) else tuple(step_kernel)
            steps = max(k["n"] for name_, k in ks.items() if any(s_ in name_ for s_ in names))
            tot = sum((k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 * k["n"] / steps for k in ks.values())
            return tot, "profiles/" + name
        except Exception:
            continue
    return None, None


def plan_exchange_bytes(plan, dim):
    """Per-step exchange volume of a planned row-sharded epoch on THIS rank (bytes, mean over the epoch's steps):
    item rows out + gradients back, all of it and the part that leaves the GPU (the rank's own share of an exchange is
    a device-to-device copy).  From the plan's host-side counts: in_cnt_h[s][q] rows go to peer q, req_cnt_h[s][q]
    rows come from owner q; every (step, peer) pair carries one extra row (the loss partials)."""
    ld, S, me = 4 * (dim + 1), plan["S"], plan["rank"]
    out_rows = sum(sum(c) + len(c) for c in plan["in_cnt_h"])
    back_rows = sum(sum(c) + len(c) for c in plan["req_cnt_h"])
    own = sum(c[me] + 1 for c in plan["in_cnt_h"]) + sum(c[me] + 1 for c in plan["req_cnt_h"])
    total = (out_rows + back_rows) * ld / S
    return {"exchange_bytes_per_step": total, "exchange_bytes_per_step_off_gpu": total - own * ld / S}


def self_exchange_fields(eng, exchange, step_s):
    """--force-exchange: the rank's own share of both exchanges went through RCCL as well (grouped ncclSend + ncclRecv to
    itself), so ALL of `exchange_bytes_per_step` crossed the communicator; at world 1 that is a measured loop-back
    rate of the real send / recv path (a device-to-device copy by RCCL's kernels, not xGMI)."""
    if not getattr(eng, "_self_exchange", False):
        return {}
    return {"self_exchange": True,
            "a2a_GBps_per_gpu": exchange["exchange_bytes_per_step"] / step_s / 1e9,
            "a2a_note": "own segment included: sent to and received from this rank itself through the real RCCL "
                        "communicator (loop-back), a lower bound like the off-GPU figure (the exchanges are two of the "
                        "step's six stages)"}


def bench_mf(args, device, world, rank, dist_on, force_mode=None, force_scaling=None, group=None):
    """BASELINE configs[1] (the headline).  N = 1: MFEngine's resident epoch (one fused launch per step).
    N > 1: row-sharded tables with planned all-to-all routing (north_star's split; `--multi-gpu sharded`, and what
    `--multi-gpu auto` reports as the headline) or replicated tables + one all-reduce per step (`replicated`)."""
    import beta_recsys_amd as hp

    K, W = args.steps, args.warmup
    scaling = force_scaling or args.scaling
    strong = scaling == "strong" and dist_on
    if strong and B % world:
        raise SystemExit(f"--scaling strong splits the batch of {B} over the ranks: {world} does not divide it")
    b_local = B // world if strong else B
    # ~1 000 000 triples per epoch = EPOCH_STEPS full batches; this rank's share is resident
    users, pos, neg = (t.to(device) for t in synth_triples(EPOCH_STEPS * b_local, seed=100 + rank))
    torch.manual_seed(7 + rank)  # keys of the device-side shuffles
    mode = "single"
    if dist_on:
        from beta_recsys_amd.replicated import ReplicatedMFEngine
        from beta_recsys_amd.sharded import ShardedMFEngine

        mode = force_mode or args.multi_gpu
        if mode == "auto":
            mode = "sharded"
        cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str=str(device), optimizer=args.optimizer,
                             lr=LR, batch_size=b_local, loss="bpr", dp_collective=args.dp_collective,
                             **({"shard_self_exchange": True} if mode == "sharded" and getattr(args, "force_exchange", False)
                                else {})),
               "system": {"run_dir": "/tmp/hiprec_bench_runs"}}
        torch.manual_seed(2020)
        with contextlib.redirect_stdout(io.StringIO()):
            eng = (ShardedMFEngine(cfg, process_group=group) if mode == "sharded"
                   else ReplicatedMFEngine(cfg, process_group=group))
        if mode != "sharded":
            eng._direct_communicator()   # (collective) created here, not inside a timed window when --warmup is 0
        else:
            eng._step_comm()             # likewise the step driver's communicator ...
            eng.prefetch_setup()         # ... and the side stream / process group of the prefetched plans
    else:
        eng = make_engine(device, args.optimizer)
        eng.fused_step = not args.two_kernel
    loader = hp.DeviceTripleBatcher(users, pos, neg, b_local)
    assert len(loader) == EPOCH_STEPS

    # Continuous training: epoch after epoch of the same loader (as TrainEngine._train drives it), enqueued in
    # pieces so that a timestamp can sit between two K-step windows.  A piece that starts an epoch takes its
    # staged arrays (shuffle + per-batch sort + layout, prefetched on the side stream during the previous
    # epoch) and starts the next epoch's staging; the piece that ends it enqueues the flush.
    state = {"pos": 0, "prepared": None, "it": None}

    def advance(n):
        while n > 0:
            take = min(n, EPOCH_STEPS - state["pos"])
            piece = (state["pos"], state["pos"] + take)
            if mode == "single":
                if piece[0] == 0:
                    state["prepared"] = eng.prepare_epoch(loader)
                eng.run_prepared_epoch(state["prepared"], sync=False, prefetch=loader, steps=piece)
            elif mode == "replicated":
                assert eng.run_resident_epoch(loader, steps=piece)
            else:   # row-sharded: route the whole epoch once (ids only), then exact-size exchanges per step
                if piece[0] == 0:
                    state["prepared"] = eng.take_plan(loader)   # prefetched during the previous epoch, if there was one
                eng.run_planned_epoch(state["prepared"], steps=piece, sync=False)
                if piece[1] == EPOCH_STEPS:
                    eng.prefetch_plan(loader)   # the whole epoch is enqueued: route the next one on the side stream
            state["pos"] = piece[1] % EPOCH_STEPS
            n -= take

    advance(W)
    torch.cuda.synchronize()
    per, wall = timed_repeats(lambda r: advance(K), K, device, dist_on, group)
    R = len(per)
    advance((EPOCH_STEPS - state["pos"]) % EPOCH_STEPS)   # finish the epoch in flight (flush) before reading back
    if mode == "sharded":
        eng.k.check_status()
    else:
        st = eng.epoch_stats()
        done = W + R * K
        assert st.step == done + (-done) % EPOCH_STEPS, (st.step, done)
        assert np.isfinite(st.loss) and 0.05 < st.loss < 1.4, st.loss

    # ---- dominant kernel: its launch period from HIP events on the stream it runs on ------------------
    probe_s = probe_steps = None
    if rank == 0:
        peng = eng if mode == "single" else make_engine(device, args.optimizer)
        fused = peng.fused_step and mode != "sharded"
        P = min(max(K, MIN_TIMED_STEPS), 2000)
        pu, pp, pn = (t.to(device) for t in synth_triples(P * B, seed=99))
        probe = stage(peng, hp.DeviceTripleBatcher(pu, pp, pn, B))
        peng.run_prepared_epoch(probe, sync=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        peng.run_prepared_epoch(probe, sync=False)
        e1.record()
        torch.cuda.synchronize()
        probe_s, probe_steps = e0.elapsed_time(e1) * 1e-3, P
        k_mean, _ = kernel_timing(peng, probe)

    parallelism = "single GPU"
    if dist_on:
        parallelism = (f"row-sharded tables over {world} GPUs (owner = row mod {world}), all-to-all routing of "
                       "triples / item rows / item gradients over RCCL"
                       if mode == "sharded" else
                       f"dp{world}: replicated 2.5 MB tables, one fused launch + one RCCL all-reduce of the dense "
                       f"gradient per step, global batch = {world} x {b_local}; collective enqueued by "
                       + ("the C epoch driver (ncclAllReduce on the engine's own communicator)"
                          if getattr(eng, "_direct_comm", None) is not None else "torch.distributed"))
    exchange = None
    if mode == "sharded":
        exchange = plan_exchange_bytes(state["prepared"], D)
    if rank != 0:
        return None
    n_params = (U + I) * (D + 1) + 1
    gs_bytes = algorithmic_bytes_per_triple(D) * B                     # gather / scatter rows of one batch
    step_bytes = gs_bytes + optimizer_sweep_bytes(args.optimizer, n_params)   # SURVEY §8(d) for THIS optimizer
    kind_id = {"sgd": 0, "adam": 1, "rmsprop": 2}[args.optimizer]
    if fused:
        dom_name = (f"mf_bpr_fused_kernel<1,{kind_id}> (gather + score + BPR grad + scatter + "
                    f"{args.optimizer} update, 1 launch/step)")
        dom_s = probe_s / (probe_steps + 1)      # P step launches + the sweep-only flush
        dom_bytes = step_bytes
        pmc_name = f"hiprec::mf_bpr_fused_kernel<1, {kind_id}, false>"
    else:
        dom_name = "mf_bpr_grad_kernel (gather + score + BPR grad + atomic scatter)"
        dom_s, dom_bytes, pmc_name = k_mean, gs_bytes, "hiprec::mf_bpr_grad_kernel<1>"
    traffic, traffic_src = traffic_from_profiles(pmc_name)
    out = {"metric": "training interactions/sec (BPR triples)", "unit": "triples/s"}
    out.update(timing_fields(per, wall, K, b_local, world))
    out.update({
        "n_gpus": world, "steps": K, "warmup": W, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "BPR-MF, MovieLens-1M-shaped synthetic (BASELINE configs[1]): 6040 users x 3706 items, "
                        f"dim 64, batch {b_local} triples/GPU, uniform users, Zipf(1.0) positives, uniform negatives",
            "optimizer": args.optimizer, "lr": LR, "loss": "bpr", "batch_per_gpu": b_local,
            "global_batch": b_local * world, "parallelism": parallelism,
            "rccl_world_size": world if dist_on else None,
            "epoch": f"{EPOCH_STEPS} steps = {EPOCH_STEPS * b_local * world} triples (SURVEY 8d C2: ~1M triples / epoch)",
            "timed_region": "continuous training, epoch after epoch; a repeat = the next K steps, with everything "
                            "that falls into them: per epoch one device shuffle + staging (sort of every batch by "
                            "item, layout; runs on a side stream during the previous epoch) and one flush launch",
        },
        "roofline": {
            "bound": "hbm", "kernel": dom_name,
            "achieved": dom_bytes / dom_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": dom_bytes / dom_s / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": dom_bytes,
            "algorithmic_bytes_note": f"SURVEY 8(d): {algorithmic_bytes_per_triple(D)} B/triple x {B} triples"
                                      + (f" + dense {args.optimizer} sweep {optimizer_sweep_bytes(args.optimizer, n_params)} B"
                                         if fused and args.optimizer != "sgd" else ""),
            "kernel_us": dom_s * 1e6, "kernel_us_source": f"HIP events around a {probe_steps}-step resident epoch",
            "gather_scatter_only": {"algorithmic_bytes_per_launch": gs_bytes, "achieved": gs_bytes / dom_s / 1e9,
                                    "frac": gs_bytes / dom_s / 1e9 / HBM_PEAK_GBS},
            "grad_only_kernel_us": k_mean * 1e6,
            "traffic": traffic, "traffic_source": traffic_src,
            # whole step, per GPU: (triples/s) x (row bytes per triple + this optimizer's sweep bytes per triple)
            "step_frac": out["value"] / world * (algorithmic_bytes_per_triple(D) + (step_bytes - gs_bytes) / b_local)
                         / (HBM_PEAK_GBS * 1e9),
        },
    })
    if exchange is not None:
        step_s = out["ms_per_step"] * 1e-3
        out["config"].update({k: round(v) for k, v in exchange.items()})
        # bytes this rank sends + receives over xGMI per step / the step time: a LOWER bound of the link rate (the
        # exchanges are two of the step's six stages)
        out["config"]["a2a_GBps_per_gpu"] = exchange["exchange_bytes_per_step_off_gpu"] / step_s / 1e9
        out["config"]["step_driver"] = eng._step_mode
        out["config"].update(self_exchange_fields(eng, exchange, step_s))
        # the sharded step has no single dominant launch: the roofline is the whole step per GPU on SURVEY 8d's bytes
        per_gpu = out["value"] / world
        moved = algorithmic_bytes_per_triple(D) + optimizer_sweep_bytes(args.optimizer, n_params // world) / b_local
        out["roofline"] = {"bound": "hbm", "kernel": "whole row-sharded step (per GPU): payload, owned-rows kernel on "
                                                      "(local users, fetched item slots), apply, optimizer",
                           "achieved": per_gpu * moved / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": per_gpu * moved / (HBM_PEAK_GBS * 1e9),
                           "algorithmic_bytes_per_launch": moved * b_local,
                           "algorithmic_bytes_note": f"SURVEY 8(d): {algorithmic_bytes_per_triple(D)} B/triple x {b_local} "
                                                     f"triples/GPU + this shard's dense {args.optimizer} sweep",
                           "kernel_us": step_s * 1e6, "kernel_us_source": "the timed step itself (HIP events)",
                           "traffic": None, "traffic_source": None}
    if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed on rank 0 at N = 1 only
        out["cpu_baseline"] = cpu_baseline(args.optimizer)
    return out


def compact_line(out, keep=("value", "unit", "ms_per_step", "scaling", "n_gpus", "steps", "repeats")):
    """A named sub-record of the one JSON line: the headline fields of another measurement of the same run."""
    if out is None:
        return None
    rec = {k: out[k] for k in keep if k in out}
    cfg = out.get("config", {})
    rec.update({k: cfg[k] for k in ("workload", "parallelism", "global_batch", "rccl_world_size", "optimizer",
                                    "exchange_bytes_per_step", "exchange_bytes_per_step_off_gpu", "a2a_GBps_per_gpu",
                                    "step_driver") if k in cfg})
    rec["roofline_frac"] = out.get("roofline", {}).get("frac")
    return rec


def bench_mf_multi_gpu(args, device, world, rank, group=None):
    """`bench.py --gpus N` with no other flag: BASELINE.json's split is the headline -- tables ROW-SHARDED over the N
    ranks (owner = row mod N), index minibatches routed to the owning rank by planned all-to-alls, at the reference's
    own batch (global batch 4096 split over the ranks: `scaling: strong`, the single-GPU step's semantics) -- and the
    two other N-GPU forms this repo has ride along as named sub-records: `alt.replicated` (data-parallel replicas of
    the 2.5 MB tables, one all-reduce per step, weak scaling) and `alt.c4_sharded` (BASELINE configs[3], 10 M x 1 M x
    128 row-sharded, 65 536 triples per GPU and step).  One JSON line."""
    import copy

    out = bench_mf(args, device, world, rank, True, force_mode="sharded", force_scaling="strong", group=group)

    def side(fn):
        # a sub-record must not cost the run its headline: an error in one (the same code and shapes run on every
        # rank, so it is raised by all of them) is reported in its place
        try:
            return compact_line(fn()) if rank == 0 else (fn() and None)
        except Exception as e:  # noqa: BLE001
            return {"error": f"{type(e).__name__}: {e}"[:400]}

    rep = side(lambda: bench_mf(args, device, world, rank, True, force_mode="replicated", force_scaling="weak", group=group))
    c4args = copy.copy(args)
    c4args.steps, c4args.warmup = min(args.steps, 50), min(args.warmup, 5)
    c4 = side(lambda: bench_mf_c4_sharded(c4args, device, world, rank, group=group))
    if rank != 0:
        return None
    out["alt"] = {"replicated": rep, "c4_sharded": c4}
    # the step every N > 1 figure of this line is to be compared with, measured in the SAME run on rank 0's GPU while the
    # other ranks wait at the final barrier: configs[1] on the single-GPU fused engine
    try:
        one = copy.copy(args)
        one.no_cpu_baseline, one.two_kernel = True, False
        with contextlib.redirect_stdout(io.StringIO()):
            single = bench_mf(one, device, 1, 0, False)
        out["single_gpu_ms_per_step"] = single["ms_per_step"]
        out["single_gpu_value"] = single["value"]
    except Exception as e:  # noqa: BLE001
        out["single_gpu_ms_per_step"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


ALT_WORKLOADS = ("ncf", "mf_c4shard_sgd", "mf_c4shard_adam", "mf_c4_adam_fullcov", "lightgcn")


def alt_single_gpu(args, device, only=None):
    """The other BASELINE.json configs under the SAME clock as the headline: the stock
    `python bench.py --gpus 1 --steps K --warmup W` keeps configs[1] as its line and carries, as `alt`, one sub-record
    per remaining config measured LIVE in this process -- `ncf` (configs[2], emb_dim 32), `mf_c4shard_sgd` /
    `mf_c4shard_adam` (one rank's share of configs[3]; SGD as owner pulls, Adam as the exact lazy form),
    `mf_c4_adam_fullcov` (configs[3] WHOLE on this GPU with the reference's default optimizer, every row met in every
    epoch) and `lightgcn` (configs[4]).  Every record: `value`, `unit`, `ms_per_step`, `steps`, `repeats`, the
    workload, and `roofline` with the byte / flop model named and `kernel_us` from HIP events of this run.  No CPU
    legs, at most 50-step windows; an error in one record is reported in its place.  `--no-alt` skips all of it."""
    import copy
    import gc

    def sub(**kw):
        a = copy.copy(args)
        a.steps, a.warmup, a.no_cpu_baseline = min(args.steps, 50), min(args.warmup, 10), True
        for k, v in kw.items():
            setattr(a, k, v)
        return a

    def record(out, model):
        roof = out.get("roofline", {})
        live_us = roof.get("kernel_us") if str(roof.get("kernel_us_source", "HIP")).startswith("HIP") else None
        rec = {k: out[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "repeats", "wall_ms_per_step") if k in out}
        rec["workload"] = out.get("config", {}).get("workload")
        for k in ("ms_per_step_kernels_alone", "ms_per_step_whole_epochs", "epoch", "rows_met_per_epoch", "last_loss"):
            if k in out.get("config", {}):
                rec[k] = out["config"][k]
        rec["roofline"] = {"bound": roof.get("bound"), "frac": roof.get("frac"), "achieved": roof.get("achieved"),
                           "peak": roof.get("peak"), "unit": roof.get("unit"), "model": model,
                           # measured in THIS run: the period of the step's launches between two HIP events
                           "kernel_us": live_us if live_us is not None else out["ms_per_step"] * 1e3,
                           "kernel_us_source": ("HIP events around an epoch staged beforehand (the step's launches alone)"
                                                if live_us is not None else "HIP events around the K-step windows"),
                           "kernel": roof.get("kernel")}
        for k in ("step_frac", "algorithmic_bytes_per_launch", "flops_per_sample", "algorithmic_bytes_per_step",
                  "dense_equivalent_GBps"):
            if k in roof:
                rec["roofline"][k] = roof[k]
        return rec

    jobs = {
        # (ncf / lightgcn: 200 untimed steps first -- the GPU comes out of the seconds of host-side set-up (graph
        # factoring, table initialisation) at idle clocks and the first ~200 LightGCN steps of a process run 124 -> 113 us,
        # profiles/r06_experiments.md 81; the record states its warm-up)
        "ncf": (lambda: bench_ncf(sub(emb_dim=32, warmup=max(args.warmup, 200)), device),
                "fwd + dgrad + wgrad flops of the tower and head per sample (SURVEY 8d: 258 432 at emb_dim 32) over the "
                "whole step, against the dense fp32 MFMA peak"),
        "mf_c4shard_sgd": (lambda: bench_mf_c4shard(sub(c4_optimizer="sgd", sgd_mode="owned"), device, full=False),
                           "SURVEY 8d: 24 + 24 (D+1) = 3 120 B per triple x 65 536 triples over the step's two launches "
                           "(frac) and over the step with the staging in the clock (step_frac)"),
        "mf_c4shard_adam": (lambda: bench_mf_c4shard(sub(c4_optimizer="adam"), device, full=False),
                            "rows of the step: 3 120 B per triple + w, m, v of its 3 rows read and written (exact lazy "
                            "Adam); SURVEY 8d's dense 28 P per step is `dense_equivalent_GBps`"),
        "mf_c4_adam_fullcov": (lambda: bench_mf_c4shard(sub(c4_optimizer="adam", epoch_coverage="full"), device, full=True),
                               "as mf_c4shard_adam, on the whole 10 M x 1 M x 128 table, every row met in every epoch; "
                               "value / ms_per_step are the median K-step window, ms_per_step_whole_epochs counts the "
                               "epoch's flush too"),
        "lightgcn": (lambda: bench_lightgcn(sub(warmup=max(args.warmup, 200)), device),
                     "SURVEY 8d: 2 L SpMMs x [nnz (4 + 4 + 1) + (N + 1) 8 + 2 N D 4] B per step over the whole step"),
    }
    alt, t_all = {}, time.perf_counter()
    for name in ALT_WORKLOADS:
        if only is not None and name not in only:
            continue
        t0 = time.perf_counter()
        try:
            fn, model = jobs[name]
            with contextlib.redirect_stdout(io.StringIO()):
                out = fn()
            alt[name] = record(out, model)
        except Exception as e:  # noqa: BLE001   a sub-record must not cost the run its headline
            alt[name] = {"error": f"{type(e).__name__}: {e}"[:400]}
        alt[name]["wall_s"] = round(time.perf_counter() - t0, 1)
        gc.collect()
        torch.cuda.empty_cache()
    alt["wall_s"] = round(time.perf_counter() - t_all, 1)
    return alt


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--optimizer", default="adam", choices=["sgd", "adam", "rmsprop"],
                    help="mf: adam is the reference's own default (configs/mf_default.json); sgd / rmsprop are the other two torch_engine.py:23-39 builds")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true",
                    help="mf at --gpus 1: skip the `alt` sub-records (the other BASELINE configs measured in the same run)")
    ap.add_argument("--alt-only", default=None,
                    help="mf at --gpus 1: comma-separated subset of the `alt` sub-records (%s)" % ", ".join(ALT_WORKLOADS))
    ap.add_argument("--two-kernel", action="store_true",
                    help="mf: gradient kernel + dense optimizer sweep per step instead of the fused one-kernel step")
    ap.add_argument("--workload", default="mf", choices=["mf", "ncf", "lightgcn", "mf-c4shard", "mf-c4", "pgmf", "t2v", "ngcf"],
                    help="mf = BASELINE configs[1] (the headline); ncf = configs[2]; mf-c4 = configs[3] "
                         "(whole on one GPU at --gpus 1, row-sharded over the ranks at --gpus N); lightgcn = configs[4]")
    ap.add_argument("--sgd-mode", default="owned", choices=["owned", "owned_atomic", "rows", "dense"],
                    help="mf-c4 / mf-c4shard: owned = rows updated in place by their owner, shared rows pulled from a "
                         "contribution buffer (2 launches, no float atomics); owned_atomic = rounds 2-4: one launch, "
                         "shared rows collect device-scope atomics (csrc/mf_owned.hip); "
                         "rows = gradient kernel into a dense buffer + touched-rows pass (round 1)")
    ap.add_argument("--emb-dim", type=int, default=32, help="ncf: 32 (tower 256-128-64-32, primary) or 64")
    ap.add_argument("--lane-slots", type=int, default=None, choices=[16, 24, 32, 48],
                    help="lightgcn: slots per lane of the column-sliced SpMM's chunks (default: the host's choice for the graph)")
    ap.add_argument("--multi-gpu", default="auto", choices=["auto", "replicated", "sharded"],
                    help="mf, N>1: auto = the row-sharded step (BASELINE.json's split: owner = row mod N, planned "
                         "all-to-alls, the reference's batch split over the ranks) as the headline with the replicated "
                         "and the configs[3] forms as `alt` sub-records; replicated / sharded = that form alone, "
                         "under --scaling")
    ap.add_argument("--c4-optimizer", default="sgd", choices=["sgd", "adam", "rmsprop"],
                    help="mf-c4 row-sharded: sgd (SURVEY 8d primary: exact scatter) or the dense optimizers (secondary)")
    ap.add_argument("--epoch-coverage", default="sample", choices=["sample", "full"],
                    help="mf-c4 / mf-c4shard: sample = 50-step epochs of uniformly drawn users (the whole table's epoch "
                         "meets 28 %% of them); full = every user and item row occurs in every epoch (the lazy "
                         "optimizers then replay the whole table's zero-gradient steps)")
    ap.add_argument("--dense-opt", default="auto", choices=["auto", "lazy", "sweep"],
                    help="mf-c4 (sharded) with Adam / RMSprop: exact lazy replay of the step's rows (csrc/lazy_opt.hip) "
                         "or the dense sweep of the whole shard every step")
    ap.add_argument("--lazy-grad", default="pull", choices=["pull", "owned", "atomic"],
                    help="mf-c4 / mf-c4shard with lazy Adam / RMSprop: pull = gradient parts through the contribution "
                         "buffer + one apply launch (round 5); owned / atomic = gradient kernel into the dense buffer + "
                         "update launch (round 4)")
    ap.add_argument("--shard-sgd", default="pull", choices=["pull", "atomic"],
                    help="mf-c4 row-sharded with plain SGD: pull = the planned step as owner pulls (two launches, no float "
                         "atomics, the partials' publish rides along); atomic = the round 2-4 step")
    ap.add_argument("--step-driver", default="c", choices=["c", "torch"],
                    help="row-sharded planned steps: c = kernels and grouped ncclSend/ncclRecv enqueued by one C call "
                         "per range of steps; torch = torch.distributed.all_to_all_single between the launches")
    ap.add_argument("--force-exchange", action="store_true",
                    help="row-sharded planned steps: every rank ALSO sends its own segment of both exchanges to itself "
                         "through the RCCL communicator (HIPREC_SHARD_EXCHANGE_SELF; bit-identical results).  At --gpus 1 "
                         "it runs the N > 1 code on one rank, so the grouped ncclSend / ncclRecv of the C step driver "
                         "execute against the real library and a2a_GBps_per_gpu is a measured loop-back rate")
    ap.add_argument("--no-plan-prefetch", action="store_true",
                    help="mf-c4 on N > 1 GPUs: plan every epoch synchronously instead of during the previous one")
    ap.add_argument("--dp-collective", default=os.environ.get("HIPREC_DP_COLLECTIVE", "rccl"), choices=["rccl", "torch"],
                    help="replicated mode: 'rccl' = the C epoch driver calls ncclAllReduce itself on the engine's own "
                         "communicator (falls back to torch.distributed if any rank cannot create one), 'torch' = "
                         "torch.distributed.all_reduce from the python step loop")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = every rank feeds a full batch (global batch N x B); strong = the reference's "
                         "batch split over the ranks (global batch = B, identical semantics to one GPU)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPUs are visible")
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    # HIPREC_BENCH_FORCE_SHARDED=1 exercises the N>1 code path on a single GPU (world size 1)
    dist_on = world > 1 or os.environ.get("HIPREC_BENCH_FORCE_SHARDED") == "1" or args.force_exchange
    json_fd = None
    if dist_on:
        import torch.distributed as dist

        # RCCL prints a version banner through C stdio on STDOUT when a communicator is created (buffered: it lands
        # after anything python printed).  The contract is ONE JSON line on rank 0's stdout: for the whole run fd 1
        # points at stderr, and the line goes to a saved copy of the real stdout at the very end.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)

        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)
        assert dist.get_world_size() == world

    if args.workload == "ncf":
        out = bench_ncf(args, device, world, rank, dist_on)
    elif args.workload == "lightgcn":
        out = bench_lightgcn(args, device, world, rank, dist_on)
    elif args.workload in ("pgmf", "t2v"):
        out = bench_siblings(args, device) if not dist_on else None
    elif args.workload == "ngcf":
        out = bench_ngcf(args, device) if not dist_on else None
    elif args.workload in ("mf-c4shard", "mf-c4"):
        out = (bench_mf_c4_sharded(args, device, world, rank) if dist_on and args.workload == "mf-c4"
               else bench_mf_c4shard(args, device, full=args.workload == "mf-c4"))
    elif dist_on and args.multi_gpu == "auto":
        out = bench_mf_multi_gpu(args, device, world, rank)
    else:
        out = bench_mf(args, device, world, rank, dist_on)
        if not dist_on and out is not None and not args.no_alt and not args.two_kernel:
            out["alt"] = alt_single_gpu(args, device, only=args.alt_only.split(",") if args.alt_only else None)
    if dist_on and args.workload in ("pgmf", "t2v", "ngcf"):
        raise SystemExit(f"--workload {args.workload} is single-GPU (no data-parallel wrapper)")
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and out is not None:
        stamp_roofline(out, {"mf": "adam", "mf-c4": "mf-c4_sharded" if dist_on else "mf-c4"}.get(args.workload, args.workload))
        if json_fd is not None:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
