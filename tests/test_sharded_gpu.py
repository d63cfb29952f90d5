"""The row-sharded engine with the REAL kernel backend (libhiprec) on one MI355X: a world-size-1
process group over RCCL runs every exchange (all-to-all with itself) and every kernel of the
sharded step; the result must equal the oracle's single-process step."""
import contextlib
import io
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from helpers import REL  # noqa: F401  (north_star: 1e-5 relative)
from helpers import KEYS, assert_mf_end_state, assert_ncf_end_state, assert_scalar_close
from oracle import mf_numpy as onp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group(hip_device):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=hip_device)
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("routing,sgd_mode", [("padded", "dense"), ("variable", "dense"), ("padded", "rows")])
@pytest.mark.parametrize("optimizer,lr", [("sgd", 0.1), ("adam", 0.05)])
def test_sharded_step_with_hip_kernels(nccl_group, optimizer, lr, routing, sgd_mode):
    from beta_recsys_amd.sharded import ShardedMFEngine

    U, I, D, B = 300, 200, 64, 512
    w0 = onp.init_params(U, I, D, seed=3)
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer=optimizer,
                         lr=lr, batch_size=B, loss="bpr", routing=routing, sgd_mode=sgd_mode),
           "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ShardedMFEngine(cfg, full_state={k: torch.from_numpy(v) for k, v in w0.items()})
    assert eng._rows_sgd == (sgd_mode == "rows" and optimizer == "sgd")   # Adam needs the dense sweep
    w = onp.copy_params(w0)
    st = onp.new_opt_state(w, optimizer)
    rng = np.random.default_rng(0)
    batches = []
    for _ in range(3):
        batch = (rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B))
        batch[1][: B // 4] = batch[1][0]
        batches.append(batch)
        loss, reg = eng.train_single_batch(tuple(torch.from_numpy(a) for a in batch))
        ref_loss, ref_reg = onp.mf_train_step(w, st, batch, "bpr", optimizer, lr)
        assert_scalar_close(loss, ref_loss, REL, "loss")
        assert_scalar_close(reg, ref_reg, REL, "reg")
    full = {k: v.cpu().numpy() for k, v in eng.gather_full_state_dict().items()}
    assert_mf_end_state(full, w0, batches, optimizer, lr, f"{routing} routing")


@pytest.mark.parametrize("optimizer,lr", [("sgd", 0.1), ("adam", 0.05)])
def test_replicated_engine_with_hip_kernels(nccl_group, optimizer, lr):
    """Replicated data-parallel mode at world size 1: all-reduce with itself, then the dense sweep."""
    from beta_recsys_amd.replicated import ReplicatedMFEngine

    U, I, D, B = 300, 200, 64, 512
    w0 = onp.init_params(U, I, D, seed=4)
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer=optimizer,
                         lr=lr, batch_size=B, loss="bpr"),
           "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ReplicatedMFEngine(cfg)
    eng.model.load_state_dict({k: torch.from_numpy(v) for k, v in w0.items()})
    w = onp.copy_params(w0)
    st = onp.new_opt_state(w, optimizer)
    rng = np.random.default_rng(1)
    batches, np_batches = [], []
    for _ in range(3):
        batch = (rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B))
        np_batches.append(batch)
        batches.append(tuple(torch.from_numpy(a) for a in batch))
        loss, reg = eng.train_single_batch(batches[-1])
        ref_loss, ref_reg = onp.mf_train_step(w, st, batch, "bpr", optimizer, lr)
        assert_scalar_close(loss, ref_loss, REL, "loss")
        assert_scalar_close(reg, ref_reg, REL, "reg")
    got = {k: v.cpu().numpy() for k, v in eng.model.state_dict().items()}
    assert_mf_end_state(got, w0, np_batches, optimizer, lr, "replicated steps")
    with contextlib.redirect_stdout(io.StringIO()):
        eng.train_an_epoch(batches, 1)
    assert len(eng.writer.scalars) == 2


def test_route_bucket_kernel(hip_device):
    """Device bucketing: every key lands in its destination's bucket exactly once, padding stays
    -1, a too-small capacity raises the overflow bit."""
    from beta_recsys_amd import _lib
    from beta_recsys_amd.sharded import HipKernels
    from beta_recsys_amd.mf import read_stats

    k = HipKernels(hip_device)
    rng = np.random.default_rng(0)
    keys = rng.integers(0, 10_000, 5000)
    keys[rng.random(5000) < 0.2] = -1
    kt = torch.from_numpy(keys).cuda()
    R, cap = 8, 700
    slots = k.route_bucket(kt, R, cap).cpu().numpy()
    live = keys >= 0
    assert np.all(slots[~live] == -1) and np.all(slots[live] >= 0)
    assert len(np.unique(slots[live])) == live.sum(), "slots must be unique"
    assert np.array_equal(slots[live] // cap, keys[live] % R), "bucket = key mod R"
    for d in range(R):
        pos = np.sort(slots[live & (keys % R == d)] - d * cap)
        assert np.array_equal(pos, np.arange(len(pos))), "buckets are filled densely from 0"
    assert read_stats(k.stats).status == 0
    k.route_bucket(kt, R, 100)
    assert read_stats(k.stats).status & _lib.STATUS_ROUTE_OVERFLOW


@pytest.mark.parametrize("optimizer,lr", [("sgd", 0.1), ("adam", 0.05), ("rmsprop", 0.01)])
def test_replicated_fused_epoch_with_hip_kernels(nccl_group, optimizer, lr):
    """The data-parallel resident epoch (one fused launch + one all-reduce of [partials | gradient]
    per step) at world size 1, two epochs in a row, against the oracle stepping the same batches; the
    loss sums it leaves in hiprec_stats are the global ones."""
    from beta_recsys_amd.replicated import ReplicatedMFEngine

    U, I, D, B = 300, 200, 64, 512
    w0 = onp.init_params(U, I, D, seed=9)
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer=optimizer,
                         lr=lr, batch_size=B, loss="bpr"),
           "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ReplicatedMFEngine(cfg)
    eng.model.load_state_dict({k: torch.from_numpy(v) for k, v in w0.items()})
    w = onp.copy_params(w0)
    st = onp.new_opt_state(w, optimizer)
    rng = np.random.default_rng(3)
    np_batches = []
    for epoch, n_steps in enumerate((3, 4)):            # odd and even: both final ping-pong positions
        total = 0.0
        eng.fused_epoch_begin()
        for s in range(n_steps):
            nb = B if s < n_steps - 1 else 77           # short last batch
            batch = (rng.integers(0, U, nb), rng.integers(0, 30, nb), rng.integers(0, I, nb))
            np_batches.append(batch)
            eng.fused_step(*(torch.from_numpy(a).cuda() for a in batch))
            loss, _ = onp.mf_train_step(w, st, batch, "bpr", optimizer, lr)
            total += loss
        eng.fused_epoch_end()
        stats = eng.epoch_stats()
        assert_scalar_close(stats.loss_sum, total, REL, f"epoch {epoch} loss sum")
        # every rotating gradient buffer is clean again
        assert all(float(b[eng._scratch.numel() // 4:].abs().max()) == 0.0 for b in eng._fe["bufs"])
    assert eng.epoch_stats().step == 7
    got = {k: v.cpu().numpy() for k, v in eng.model.state_dict().items()}
    assert_mf_end_state(got, w0, np_batches, optimizer, lr, "replicated fused epochs")
    # the per-batch path still works afterwards and continues from the same state
    batch = tuple(torch.from_numpy(rng.integers(0, n, B)) for n in (U, I, I))
    loss, _ = eng.train_single_batch(batch)
    ref_loss, _ = onp.mf_train_step(w, st, tuple(t.numpy() for t in batch), "bpr", optimizer, lr)
    assert_scalar_close(loss, ref_loss, REL, "loss after the fused epochs")


def test_replicated_train_an_epoch_takes_the_fused_path_for_resident_loaders(nccl_group):
    import beta_recsys_amd as hp
    from beta_recsys_amd.replicated import ReplicatedMFEngine

    U, I, D, B = 300, 200, 32, 128
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer="adam", lr=0.02,
                         batch_size=B, loss="bpr"), "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ReplicatedMFEngine(cfg)
    rng = np.random.default_rng(2)
    n = 5 * B + 40
    loader = hp.DeviceTripleBatcher(*(torch.from_numpy(rng.integers(0, m, n)).cuda() for m in (U, I, I)), B)
    sums = []
    with contextlib.redirect_stdout(io.StringIO()):
        for epoch in range(3):
            eng.train_an_epoch(loader, epoch)
            sums.append(eng.writer.scalars[-2][1])
    assert eng._fe is not None and eng.epoch_stats().step == 18       # 6 fused steps per epoch
    assert all(np.isfinite(sums)) and sums[2] < sums[0]


@pytest.mark.parametrize("optimizer", ["adam", "sgd"])
def test_replicated_epoch_driver_calls_rccl_itself(nccl_group, optimizer):
    """The resident data-parallel epoch is ONE C call per piece (hiprec_mf_bpr_dp_epoch_fused_range) that enqueues the
    fused launch and ncclAllReduce of every step itself, on a communicator the engine creates with _rccl; it must
    leave the state the torch.distributed loop leaves (``dp_collective: "torch"``), whole epochs and in pieces."""
    import beta_recsys_amd as hp
    from beta_recsys_amd.replicated import ReplicatedMFEngine

    U, I, D, B = 300, 200, 32, 128
    rng = np.random.default_rng(2)
    n = 5 * B + 40
    data = [torch.from_numpy(rng.integers(0, m, n)).cuda() for m in (U, I, I)]
    out = {}
    for mode in ("rccl", "torch"):
        cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer=optimizer, lr=0.02,
                             batch_size=B, loss="bpr", dp_collective=mode, shuffle_seed=7),
               "system": {"run_dir": "/tmp/hiprec_test_runs"}}
        torch.manual_seed(1)
        with contextlib.redirect_stdout(io.StringIO()):
            eng = ReplicatedMFEngine(cfg)
        loader = hp.DeviceTripleBatcher(*data, B)
        w_start = {k: v.cpu().numpy().copy() for k, v in eng.model.state_dict().items()}
        # the batches the three epochs will visit: the engine shuffles with seeds drawn from torch's CPU generator
        # (DeviceTripleBatcher.draw_seed) -- replay the draws, then rewind the generator
        rng_state = torch.get_rng_state()
        visited = []
        for _ in range(3):
            perm = loader.permutation().cpu().numpy()
            visited += [tuple(d.cpu().numpy()[perm[k:k + B]] for d in data) for k in range(0, n, B)]
        torch.set_rng_state(rng_state)
        sums = []
        with contextlib.redirect_stdout(io.StringIO()):
            for epoch in range(2):
                eng.train_an_epoch(loader, epoch)
                sums.append(eng.writer.scalars[-2][1])
            # a third epoch enqueued in two pieces
            assert eng.run_resident_epoch(loader, steps=(0, 2)) and eng.run_resident_epoch(loader, steps=(2, 6))
            sums.append(eng._sync_stats().loss_sum)
        assert (eng._direct_comm is not None) == (mode == "rccl")
        assert eng.epoch_stats().step == 18
        out[mode] = (sums, {k: v.cpu().numpy() for k, v in eng.model.state_dict().items()}, w_start, visited)
    for a, b in zip(out["rccl"][0], out["torch"][0]):
        assert_scalar_close(a, b, REL, "epoch loss sums of the two drivers")
    # both drivers ran the same three epochs (same shuffle seeds): both end states lie on the oracle's trajectory
    for mode in ("rccl", "torch"):
        assert_mf_end_state(out[mode][1], out[mode][2], out[mode][3], optimizer, 0.02, f"{mode} driver")


def test_replicated_ncf_engine_with_hip_kernels(nccl_group):
    """Data-parallel NeuMF at world size 1: gradient kernel with the global 1/B, all-reduce of
    [gradient | loss], dense Adam sweep; losses and weights follow the oracle."""
    from oracle import ncf_numpy as onc

    import beta_recsys_amd as hp
    from beta_recsys_amd.replicated import replicated_ncf_engine

    U, I, E, L, B = 120, 90, 32, 3, 200
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=E, dropout=0.0, device_str="cuda:0", optimizer="adam",
                         lr=1e-3, batch_size=B, model="ncf_end", mlp_config={"n_layers": L}, gmf_config={}),
           "system": {"run_dir": "/tmp/hiprec_test_runs", "model_save_dir": "/tmp/hiprec_test_runs"}}
    torch.manual_seed(4)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = replicated_ncf_engine(hp.NeuMFEngine)(cfg)
    w = {k: v.detach().cpu().numpy().copy() for k, v in eng.model.state_dict().items()}
    w_start = {k: v.copy() for k, v in w.items()}
    st = onc.new_opt_state(w, "adam")
    rng = np.random.default_rng(4)
    batches = []
    for _ in range(3):
        users, items = rng.integers(0, U, B), rng.integers(0, I, B)
        ratings = (rng.random(B) < 0.3).astype(np.float32)
        batches.append((users, items, ratings))
        loss = eng.train_single_batch(torch.from_numpy(users), torch.from_numpy(items), torch.from_numpy(ratings))
        ref = onc.ncf_train_step(w, st, (users, items, ratings), "neumf", "adam", 1e-3)
        assert_scalar_close(loss, ref, REL, "loss")
    got = {k: v.cpu().numpy() for k, v in eng.model.state_dict().items()}
    assert_ncf_end_state(got, w_start, batches, "neumf", "adam", 1e-3, "replicated NeuMF")


@pytest.mark.parametrize("kind,emb", [("neumf", 32), ("neumf", 8), ("gmf", 16), ("mlp", 16)])
def test_sharded_ncf_engine_with_hip_kernels(nccl_group, kind, emb):
    """sharded_ncf.py with the real kernels at world size 1 (RCCL all-to-all / all-reduce with itself): ids routed
    to the owner, rows fetched, hiprec_ncf_grad on the fetched buffers, row gradients returned and scattered, dense
    Adam over [table shards | tower | head] -- equal to oracle/ncf_numpy.py's single-process steps."""
    import beta_recsys_amd as hp
    from beta_recsys_amd import sharded_ncf
    from oracle import ncf_numpy as onc

    U, I, L, B = 300, 200, 3, 512
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=emb, dropout=0.0, device_str="cuda:0", optimizer="adam",
                         lr=0.01, batch_size=B, model="ncf_end", mlp_config={"n_layers": L}, gmf_config={}),
           "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    torch.manual_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        full = getattr(hp, {"neumf": "NeuMF", "gmf": "GMF", "mlp": "MLP"}[kind])(dict(cfg["model"], device_str="cpu")).state_dict()
    cls = {"neumf": sharded_ncf.ShardedNeuMFEngine, "gmf": sharded_ncf.ShardedGMFEngine,
           "mlp": sharded_ncf.ShardedMLPEngine}[kind]
    eng = cls(cfg, full_state=full)
    w = {k: v.numpy().copy() for k, v in full.items()}
    w_start = {k: v.copy() for k, v in w.items()}
    st = onc.new_opt_state(w, "adam")
    rng = np.random.default_rng(1)
    batches = []
    for _ in range(3):
        users, items = rng.integers(0, U, B), rng.integers(0, I, B)
        items[: B // 4] = items[0]
        ratings = (rng.random(B) < 0.2).astype(np.float32)
        batches.append((users, items, ratings))
        loss = eng.train_single_batch(users, items, ratings)
        ref = onc.ncf_train_step(w, st, (users, items, ratings), kind, "adam", 0.01)
        assert_scalar_close(loss, ref, REL, "loss")
    out = {k: v.cpu().numpy() for k, v in eng.gather_full_state_dict().items()}
    assert set(out) == set(w) and all(out[k].shape == w[k].shape for k in w)
    assert_ncf_end_state(out, w_start, batches, kind, "adam", 0.01, f"sharded {kind}")
    with pytest.raises(IndexError):
        eng.train_single_batch(np.array([U]), np.array([0]), np.array([1.0], dtype=np.float32))


@pytest.mark.parametrize("D,B,shuffle,optimizer,lr", [(64, 512, False, "sgd", 0.05), (128, 1000, True, "sgd", 0.05),
                                                      (100, 300, False, "sgd", 0.05), (64, 512, True, "adam", 0.05),
                                                      (128, 300, False, "rmsprop", 0.01), (2, 200, False, "sgd", 0.05)])
@pytest.mark.parametrize("driver", ["c", "torch"])
def test_planned_sharded_epoch_with_hip_kernels(nccl_group, D, B, shuffle, optimizer, lr, driver):
    """ShardedMFEngine.train_an_epoch on a device-resident loader: epoch-level routing by the planner kernels, then
    per step payload -> exchange -> gradient kernel on (local users, fetched item slots) -> partials into the extra
    rows -> exchange -> apply (-> dense sweep for Adam / RMSprop), all with the real kernels at world size 1 -- through
    the C step driver (one call per epoch, no exchange at world 1) and through the torch.distributed loop; equal to
    the oracle's single-process steps: SGD every element within 1e-5 of the update, the dense optimizers every
    element inside the legal-trajectory envelope."""
    import beta_recsys_amd as hp
    from beta_recsys_amd.sharded import ShardedMFEngine
    from helpers import assert_on_trajectory, assert_sgd_exact, mf_trajectory

    U, I, n = 3000, 400, 4 * B + B // 3
    w0 = onp.init_params(U, I, D, seed=3)
    rng = np.random.default_rng(D)
    p = 1.0 / np.arange(1, I + 1)
    users, pos, neg = rng.integers(0, U, n), rng.choice(I, n, p=p / p.sum()), rng.integers(0, I, n)
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer=optimizer, lr=lr, batch_size=B,
                         loss="bpr", sgd_mode="rows", step_driver=driver), "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ShardedMFEngine(cfg, full_state={k: torch.from_numpy(v) for k, v in w0.items()})
    gen = torch.Generator().manual_seed(4) if shuffle else None
    loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a).cuda() for a in (users, pos, neg)), B, shuffle=shuffle,
                                    generator=gen)
    with contextlib.redirect_stdout(io.StringIO()):
        total_loss, total_reg = eng.train_an_epoch(loader, 0)
    assert eng._step_mode == driver
    order = torch.randperm(n, generator=torch.Generator().manual_seed(4)).numpy() if shuffle else np.arange(n)
    batches = [(users[order[k:k + B]], pos[order[k:k + B]], neg[order[k:k + B]]) for k in range(0, n, B)]
    w = onp.copy_params(w0)
    st = onp.new_opt_state(w, optimizer)
    ref_loss = ref_reg = 0.0
    for batch in batches:
        loss, reg = onp.mf_train_step(w, st, batch, "bpr", optimizer, lr)
        ref_loss += loss
        ref_reg += reg
    assert_scalar_close(total_loss, ref_loss, REL, "epoch loss sum")
    assert_scalar_close(total_reg, ref_reg, REL, "epoch regularizer sum")
    full = {k: v.cpu().numpy() for k, v in eng.gather_full_state_dict().items()}
    if optimizer == "sgd":
        assert_sgd_exact(full, w, w0, "planned epoch", lr=lr, batch=B)
        assert float(eng._planned_bufs["acc"].abs().max()) == 0.0 and int(eng._planned_bufs["arrived"].abs().max()) == 0
    else:
        w_ref, env, upd = mf_trajectory(w0, batches, optimizer, lr)
        assert_on_trajectory(full, w_ref, env, upd, "planned epoch")
        assert float(eng._g_flat.abs().max()) == 0.0, "the sweep leaves the dense gradient zeroed"


def test_planned_epoch_raises_index_error_for_out_of_range_ids(nccl_group):
    """the epoch plan validates user / item ids (IndexError, like nn.Embedding), it never aliases them."""
    import beta_recsys_amd as hp
    from beta_recsys_amd.sharded import ShardedMFEngine

    U, I, D, B = 300, 40, 8, 64
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer="sgd", lr=0.05, batch_size=B,
                         loss="bpr"), "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ShardedMFEngine(cfg)
    rng = np.random.default_rng(0)
    users, pos, neg = rng.integers(0, U, 200), rng.integers(0, I, 200), rng.integers(0, I, 200)
    pos[77] = I
    loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a).cuda() for a in (users, pos, neg)), B, shuffle=False)
    with pytest.raises(IndexError):
        eng.train_an_epoch(loader, 0)
    pos[77] = 0
    loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a).cuda() for a in (users, pos, neg)), B, shuffle=False)
    with contextlib.redirect_stdout(io.StringIO()):
        eng.train_an_epoch(loader, 0)    # the engine is usable afterwards


def test_planned_epochs_with_the_plan_prefetched_on_a_side_stream(nccl_group):
    """prefetch_plan routes the NEXT epoch on a side stream / its own process group while the current one runs;
    take_plan hands it over behind an event.  Four shuffled epochs that way equal four planned synchronously."""
    import beta_recsys_amd as hp
    from beta_recsys_amd.sharded import ShardedMFEngine

    U, I, D, B = 3000, 400, 64, 512
    n = 6 * B + 100
    w0 = onp.init_params(U, I, D, seed=3)
    rng = np.random.default_rng(1)
    p = 1.0 / np.arange(1, I + 1)
    data = [torch.from_numpy(a).cuda() for a in (rng.integers(0, U, n), rng.choice(I, n, p=p / p.sum()),
                                                  rng.integers(0, I, n))]
    results = {}
    for mode in ("prefetch", "sync"):
        cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer="sgd", lr=0.05,
                             batch_size=B, loss="bpr"), "system": {"run_dir": "/tmp/hiprec_test_runs"}}
        with contextlib.redirect_stdout(io.StringIO()):
            eng = ShardedMFEngine(cfg, full_state={k: torch.from_numpy(v) for k, v in w0.items()})
        loader = hp.DeviceTripleBatcher(*data, B, shuffle=True, generator=torch.Generator().manual_seed(4))
        sums = []
        for epoch in range(4):
            plan = eng.take_plan(loader) if mode == "prefetch" else eng.plan_epoch(loader)
            assert ("ready" in plan) == (mode == "prefetch" and epoch > 0)
            eng.run_planned_epoch(plan, sync=False)
            if mode == "prefetch" and epoch < 3:
                eng.prefetch_plan(loader)          # while the epoch just enqueued runs
            sums.append(eng.k.epoch_stats()[2])
        results[mode] = (sums, {k: v.cpu().numpy() for k, v in eng.gather_full_state_dict().items()})
    for a, b in zip(*[results[m][0] for m in ("prefetch", "sync")]):
        assert_scalar_close(a, b, REL, "epoch loss sums")
    for k in KEYS:
        a, b = results["prefetch"][1][k], results["sync"][1][k]
        assert np.abs(a - b).max() <= 1e-5 * max(np.abs(b - w0[k]).max(), 1e-6) + 4 * 1.2e-7 * np.abs(b).max(), k


def test_planned_sharded_epoch_at_c4_shard_size(nccl_group):
    """The sharded engine's step at one rank's share of BASELINE configs[3] (1.25M x 125k rows, dim 128, 65536
    triples per step) in -m gpu: at world size 1 the row-sharded engine and the single-GPU engine run
    the same SGD steps -- equal loss sums and weights (1e-5 of the update), rows outside the batches bit-identical."""
    import beta_recsys_amd as hp
    from beta_recsys_amd.sharded import ShardedMFEngine

    U, I, D, B, steps = 1_250_000, 125_000, 128, 65536, 3
    rng = np.random.default_rng(11)
    pz = 1.0 / np.arange(1, I + 1)
    users, pos, neg = rng.integers(0, U, steps * B), rng.permutation(I)[rng.choice(I, steps * B, p=pz / pz.sum())], rng.integers(0, I, steps * B)
    triples = [torch.from_numpy(a).cuda() for a in (users, pos, neg)]
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer="sgd", lr=0.05, batch_size=B,
                         loss="bpr", sgd_mode="owned"), "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    torch.manual_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        single = hp.MFEngine(cfg)
    w0 = single.model.flat.clone()
    torch.manual_seed(5)   # the same seed -> the same initial model (drawn in full, every rank keeps its rows)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ShardedMFEngine(cfg)
    assert torch.equal(eng.model.flat, w0)
    loader = hp.DeviceTripleBatcher(*triples, B, shuffle=False)
    with contextlib.redirect_stdout(io.StringIO()):
        total_loss, _ = eng.train_an_epoch(loader, 0)
        single.train_an_epoch(loader, 0)
    ref = single.model.flat
    assert_scalar_close(total_loss, single.epoch_stats().loss_sum, 1e-5, "epoch loss sum, sharded vs single GPU")
    upd = float((ref - w0).abs().max())
    assert float((eng.model.flat - ref).abs().max()) <= 1e-5 * upd + 4 * 1.2e-7 * float(w0.abs().max())
    # rows no triple of the epoch names are bit-identical to the initial model in BOTH engines (plain SGD moves
    # touched rows only); inside touched rows the two engines sum in different orders, so "moved / not moved" of a
    # single element may differ by an ulp-sized update and is not compared
    m = eng.model
    for name, ids, n_rows in (("user", users, U), ("item", np.concatenate([pos, neg]), I)):
        idle = torch.ones(n_rows, dtype=torch.bool, device="cuda")
        idle[torch.from_numpy(np.unique(ids)).cuda()] = False
        ue, ie, ub, ib, _ = m._views(m.flat)
        ue0, ie0, ub0, ib0, _ = m._views(w0)
        re_, rie, rub, rib, _ = m._views(ref)
        emb, emb0, remb = (ue, ue0, re_) if name == "user" else (ie, ie0, rie)
        bias, bias0, rbias = (ub, ub0, rub) if name == "user" else (ib, ib0, rib)
        assert idle.any()
        assert torch.equal(emb[idle], emb0[idle]) and torch.equal(bias[idle], bias0[idle]), f"idle {name} rows moved"
        assert torch.equal(remb[idle], emb0[idle]) and torch.equal(rbias[idle], bias0[idle])


def test_planned_sharded_epoch_on_the_whole_configs3_table(nccl_group):
    """BASELINE configs[3] at FULL size in -m gpu: 10 M users x 1 M items x dim 128 (5.7 GB of
    tables) on the row-sharded engine at world size 1, two planned steps of 65 536 triples (uniform users, Zipf
    positives).  Checked against the numpy oracle run on the COMPACTED problem -- the rows the two batches touch,
    renumbered; SGD never reads or writes any other row -- loss sums to 1e-5, every touched element within 1e-5 of
    the update; and size-independent properties: every untouched row bit-identical, accumulators clean."""
    import beta_recsys_amd as hp
    from beta_recsys_amd.sharded import ShardedMFEngine
    from helpers import assert_sgd_exact

    U, I, D, B, steps, lr = 10_000_000, 1_000_000, 128, 65536, 2, 0.05
    rng = np.random.default_rng(12)
    pz = 1.0 / np.arange(1, I + 1)
    users = rng.integers(0, U, steps * B)
    pos = rng.permutation(I)[rng.choice(I, steps * B, p=pz / pz.sum())]
    neg = rng.integers(0, I, steps * B)
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer="sgd", lr=lr, batch_size=B,
                         loss="bpr", shard_init="local"), "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    torch.manual_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ShardedMFEngine(cfg)
    m = eng.model
    w0 = m.flat.clone()
    loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a).cuda() for a in (users, pos, neg)), B, shuffle=False)
    with contextlib.redirect_stdout(io.StringIO()):
        total_loss, total_reg = eng.train_an_epoch(loader, 0)
    assert eng._step_mode == "c"
    # the compacted problem
    uu, u_inv = np.unique(users, return_inverse=True)
    ui, i_inv = np.unique(np.concatenate([pos, neg]), return_inverse=True)
    tu, ti = torch.from_numpy(uu).cuda(), torch.from_numpy(ui).cuda()
    ue0, ie0, ub0, ib0, gb0 = m._views(w0)
    wc0 = {"user_emb.weight": ue0[tu].cpu().numpy(), "item_emb.weight": ie0[ti].cpu().numpy(),
           "user_bias.weight": ub0[tu].cpu().numpy(), "item_bias.weight": ib0[ti].cpu().numpy(),
           "global_bias": gb0.cpu().numpy().copy()}
    w = onp.copy_params(wc0)
    st = onp.new_opt_state(w, "sgd")
    ref_loss = ref_reg = 0.0
    for k in range(steps):
        sl = slice(k * B, (k + 1) * B)
        loss, reg = onp.mf_train_step(w, st, (u_inv[sl], i_inv[:steps * B][sl], i_inv[steps * B:][sl]), "bpr", "sgd", lr)
        ref_loss += loss
        ref_reg += reg
    assert_scalar_close(total_loss, ref_loss, 1e-5, "epoch loss sum vs the oracle on the compacted problem")
    assert_scalar_close(total_reg, ref_reg, REL, "epoch regularizer sum")
    ue, ie, ub, ib, gb = m._views(m.flat)
    got = {"user_emb.weight": ue[tu].cpu().numpy(), "item_emb.weight": ie[ti].cpu().numpy(),
           "user_bias.weight": ub[tu].cpu().numpy(), "item_bias.weight": ib[ti].cpu().numpy(),
           "global_bias": gb.cpu().numpy()}
    assert_sgd_exact(got, w, wc0, "configs[3] full size, touched rows", lr=lr, batch=B)
    # untouched rows: bit-identical
    for emb, emb0, bias, bias0, ids, n_rows in ((ue, ue0, ub, ub0, tu, U), (ie, ie0, ib, ib0, ti, I)):
        idle = torch.ones(n_rows, dtype=torch.bool, device="cuda")
        idle[ids] = False
        assert int(idle.sum()) > n_rows // 2
        assert torch.equal(emb[idle], emb0[idle]) and torch.equal(bias[idle], bias0[idle]), "an idle row moved"
    assert float(eng._planned_bufs["acc"].abs().max()) == 0.0 and int(eng._planned_bufs["arrived"].abs().max()) == 0


@pytest.mark.parametrize("model", ["lightgcn", "ngcf", "pgmf", "t2v"])
@pytest.mark.parametrize("collective", ["rccl", "torch"])
def test_replicated_flat_engines_with_hip_kernels(nccl_group, model, collective):
    """replicated.replicated_flat_engine (data-parallel LightGCN / NGCF / PairwiseGMF / Triple2vec) with the real
    kernels at world size 1: gradient kernel with the share-scaled batch mean, loss share into the tail, ONE
    all-reduce of [gradient | loss] (direct ncclAllReduce on the engine's own communicator, or torch.distributed),
    PairwiseGMF's clip AFTER the reduction, dense sweep -- three Adam steps must follow the oracle's trajectory."""
    import beta_recsys_amd as hp
    from beta_recsys_amd.replicated import replicated_flat_engine
    from helpers import assert_on_trajectory, oracle_trajectory
    from oracle import lightgcn_numpy as olg

    rng = np.random.default_rng(8)
    U, I, D, B, lr = 300, 200, 32, 128, 0.01
    sysd = {"run_dir": "/tmp/hiprec_test_runs"}
    adj = olg.build_norm_adj(U, I, rng.integers(0, U, 3000), rng.integers(0, I, 3000))
    co = adj.tocoo()
    tadj = torch.sparse_coo_tensor(torch.from_numpy(np.vstack((co.row, co.col)).astype(np.int64)),
                                   torch.from_numpy(co.data.astype(np.float32)), torch.Size(co.shape))
    triples = [(rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)) for _ in range(3)]
    torch.manual_seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        if model == "lightgcn":
            cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, layer_size=[D] * 2, keep_pro=1.0, regs=[1e-4],
                                 device_str="cuda:0", optimizer="adam", lr=lr, batch_size=B, norm_adj=tadj,
                                 dp_collective=collective), "system": sysd}
            eng = replicated_flat_engine(hp.LightGCNEngine)(cfg)
            eng.model.eval()     # no edge dropout: the oracle below propagates over the full graph
            grads = lambda w, b: olg.lightgcn_grads(w, adj, 2, b[0], b[1], b[2], 1e-4)[1]  # noqa: E731
            batches = triples
        elif model == "ngcf":
            from oracle import ngcf_numpy as ong

            cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, layer_size=[D, D], mess_dropout=[0.0, 0.0], regs=[1e-4],
                                 device_str="cuda:0", optimizer="adam", lr=lr, batch_size=B, norm_adj=tadj,
                                 dp_collective=collective), "system": sysd}
            eng = replicated_flat_engine(hp.NGCFEngine)(cfg)
            grads = lambda w, b: ong.ngcf_grads(w, adj, b[0], b[1], b[2], 1e-4, B)[1]  # noqa: E731
            batches = triples
        elif model == "pgmf":
            from oracle import pgmf_numpy as opg

            cfg = {"n_users": U, "n_items": I, "emb_dim": D, "regs": [1e-5], "batch_size": B, "lr": lr,
                   "pretrain_l2_lambda": 1e-2, "grad_clip": 0.05, "neg_count": 4,
                   "model": {"device_str": "cuda:0", "optimizer": "adam", "lr": lr, "dp_collective": collective},
                   "system": sysd}
            eng = replicated_flat_engine(hp.PairwiseGMFEngine)(cfg)
            grads = lambda w, b: opg.pgmf_grads(w, b[0], b[1], b[2], 1e-2)[1]  # noqa: E731
            batches = triples
        else:
            from oracle import triple2vec_numpy as ot2

            n_neg = 3
            cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, n_neg=n_neg, batch_size=B, device_str="cuda:0",
                                 optimizer="adam", lr=lr, dp_collective=collective), "system": sysd}
            eng = replicated_flat_engine(hp.Triple2vecEngine)(cfg)
            batches = [(u, p, q, rng.integers(0, U, (B, n_neg)), rng.integers(0, I, (B, n_neg)),
                        rng.integers(0, I, (B, n_neg))) for u, p, q in triples]
            grads = lambda w, b: ot2.t2v_grads(w, b, B)[1]  # noqa: E731
    w0 = {k: v.detach().cpu().numpy().copy() for k, v in eng.model.state_dict().items()}
    if model == "pgmf":        # the reference's 0.01-sized init keeps every ReLU closed: scale the memories up
        for k in ("user_memory.weight", "item_memory.weight"):
            w0[k] = w0[k] * 50.0
        eng.model.load_state_dict({k: torch.from_numpy(v) for k, v in w0.items()})
    for b in batches:
        loss = eng.train_single_batch(tuple(torch.from_numpy(np.asarray(x)) for x in b))
        assert np.isfinite(loss if not isinstance(loss, tuple) else loss[0])
    assert (eng._dar.comm is not None) == (collective == "rccl")
    from oracle import mf_numpy as omf

    def step(w, g, st):
        if model == "pgmf":
            from oracle import pgmf_numpy as opg
            opg.clip_grad_norm(g, 0.05)
        omf.opt_step(w, g, st, "adam", lr)
        if model == "t2v":
            w["item_emb2.weight"][...] = w["item_emb1.weight"]

    w_ref, env, upd = oracle_trajectory(w0, batches, grads, step, lambda w: omf.new_opt_state(w, "adam"))
    got = {k: v.detach().cpu().numpy() for k, v in eng.model.state_dict().items()}
    # NGCF: three Adam steps through two ill-conditioned hops -- which ELEMENT of a 32 x 32 hop matrix drifts furthest is
    # itself chaotic, and an elementwise maximum over eight perturbed runs under-covers the ninth (one element 2 % over
    # its own envelope in one of five GPU runs, r06): the tensor's pooled envelope is the bound there
    assert_on_trajectory(got, w_ref, env, upd, f"replicated {model}", pool=(model == "ngcf"))


def _zipf_epoch(rng, U, I, B, steps, unique=False):
    if unique:      # no row twice in a batch: a gradient element receives one add, the run is reproducible to the bit
        users = np.concatenate([rng.permutation(U)[:B] for _ in range(steps)])
        items = [rng.permutation(I)[: 2 * B] for _ in range(steps)]
        return users, np.concatenate([i[:B] for i in items]), np.concatenate([i[B:] for i in items])
    p = 1.0 / np.arange(1, I + 1)
    return (rng.integers(0, U, steps * B), rng.permutation(I)[rng.choice(I, steps * B, p=p / p.sum())],
            rng.integers(0, I, steps * B))


@pytest.mark.parametrize("optimizer,dense_opt,unique", [("sgd", "auto", True), ("sgd", "auto", False), ("adam", "lazy", True),
                                                        ("adam", "sweep", True)])
def test_self_exchange_executes_the_real_rccl_send_recv_path(nccl_group, optimizer, dense_opt, unique):
    """at world size 1 the C step driver posts no exchange, so its grouped ncclSend / ncclRecv had only
    ever run against the loopback stand-in.  ``shard_self_exchange`` (HIPREC_SHARD_EXCHANGE_SELF) makes the rank send
    its own segment of BOTH exchanges of every step to itself on the real communicator of ``_rccl.create_communicator``:
    the binding, the argument marshalling and the stream ordering run against librccl.  An exchange is a copy, so the
    epoch must equal the no-exchange run bit for bit -- weights, moments, loss sums -- wherever a run is reproducible
    to the bit in the first place (batches without a repeated row); with Zipf items the planner lays shared rows'
    contributions out in hash-table arrival order, two runs of ANY form differ in the last bits of a sum, and the two
    forms are held to 1e-5 of the update like every other pair of forms."""
    import beta_recsys_amd as hp
    from beta_recsys_amd.sharded import ShardedMFEngine

    U, I, D, B, steps = 3000, 1200, 64, 256, 5
    users, pos, neg = _zipf_epoch(np.random.default_rng(21), U, I, B, steps, unique)
    w0 = {k: torch.from_numpy(v) for k, v in onp.init_params(U, I, D, seed=7).items()}
    triples = tuple(torch.from_numpy(a).cuda() for a in (users, pos, neg))
    out = {}
    for self_x in (False, True):
        cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer=optimizer, lr=0.05,
                             batch_size=B, loss="bpr", sgd_mode="rows", dense_opt=dense_opt, shard_self_exchange=self_x),
               "system": {"run_dir": "/tmp/hiprec_test_runs"}}
        with contextlib.redirect_stdout(io.StringIO()):
            eng = ShardedMFEngine(cfg, full_state=w0)
            sums = eng.train_an_epoch(hp.DeviceTripleBatcher(*triples, B, shuffle=False), 0)
        assert eng._step_mode == "c" and eng._self_exchange == self_x
        # the communicator exists exactly when exchanges are posted, and it is RCCL's, not a test double's
        assert (eng._comm is not None) == self_x
        if self_x:
            from beta_recsys_amd import _rccl

            assert isinstance(eng._comm, _rccl.Communicator) and eng._comm.has_send_recv() and eng._comm.async_error() in (0, None)
        eng.flush_lazy()
        opt = eng.optimizer
        out[self_x] = (eng.model.flat.clone(), sums, [t.clone() for t in (opt.exp_avg, opt.exp_avg_sq) if t is not None])
        if self_x:
            eng._comm.destroy()
    (wa, sa, ma), (wb, sb, mb) = out[False], out[True]
    if unique:
        # (the epoch's loss / regularizer sums add the triples in the plan's layout order, which hash-table arrival
        # decides: last-bit differences between ANY two runs; the weights and moments are what must not move)
        for x, y in zip(sa, sb):
            assert_scalar_close(y, x, 1e-6, "epoch sums, self-exchange vs in place")
        assert torch.equal(wa, wb), f"{int((wa != wb).sum())} weights differ between the in-place and the RCCL self-exchange"
        for x, y in zip(ma, mb):
            assert torch.equal(x, y)
    else:
        flat0 = torch.cat([w0[k].reshape(-1) for k in ("user_emb.weight", "item_emb.weight", "user_bias.weight",
                                                       "item_bias.weight", "global_bias")]).cuda()
        upd = float((wa - flat0).abs().max())
        assert_scalar_close(sb[0], sa[0], 1e-5, "epoch loss sum, self-exchange vs in place")
        assert float((wa - wb).abs().max()) <= 1e-5 * upd + 4 * 1.2e-7 * float(flat0.abs().max())
    # ... and the run is the oracle's (the plain-SGD case; the Adam cases are held to it by the tests above)
    if optimizer == "sgd":
        w = onp.copy_params({k: v.numpy() for k, v in w0.items()})
        st = onp.new_opt_state(w, "sgd")
        total = sum(onp.mf_train_step(w, st, (users[k:k + B], pos[k:k + B], neg[k:k + B]), "bpr", "sgd", 0.05)[0]
                    for k in range(0, steps * B, B))
        assert_scalar_close(sb[0], total, 1e-5, "epoch loss sum vs the oracle")


def test_self_exchange_needs_the_c_driver(nccl_group):
    from beta_recsys_amd.sharded import ShardedMFEngine

    cfg = {"model": dict(n_users=50, n_items=40, emb_dim=8, device_str="cuda:0", optimizer="sgd", lr=0.05, batch_size=16,
                         loss="bpr", shard_self_exchange=True, step_driver="torch"),
           "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ShardedMFEngine(cfg)
    with pytest.raises(ValueError, match="C step driver"):
        eng._step_comm()
