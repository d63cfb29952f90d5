"""SURVEY.md 8f-3, the optional half: the optimizer state on disk and an exact resume.

The reference saves ``model.state_dict()`` only (beta_rec/models/torch_engine.py:70-90); that single file stays what
``save_checkpoint(path)`` writes.  ``save_checkpoint(path, optimizer_state=True)`` adds a second file with the optimizer
clock as the device holds it, the moments and -- for the exact lazy Adam / RMSprop form (csrc/lazy_opt.hip) -- the
per-row stamps and the per-step scalars table; ``resume_checkpoint(path, optimizer_state=True)`` restores it.  The
bar: train(2 epochs) == train(1 epoch) + save + a NEW engine + resume + train(1 epoch), bit for bit, weights and
moments.  (Batches without a repeated row: a gradient element then receives exactly one add, so a run is reproducible
to the bit in the first place.)"""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from oracle import mf_numpy as onp

pytestmark = pytest.mark.gpu


def _unique_row_batches(rng, U, I, B, steps):
    users, pos, neg = [], [], []
    for _ in range(steps):
        users.append(rng.permutation(U)[:B])
        items = rng.permutation(I)[: 2 * B]
        pos.append(items[:B])
        neg.append(items[B:])
    return tuple(np.concatenate(a).astype(np.int64) for a in (users, pos, neg))


def _engine(U, I, D, B, optimizer, **extra):
    import beta_recsys_amd as hp

    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer=optimizer, lr=0.02,
                         batch_size=B, loss="bpr", **extra), "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    with contextlib.redirect_stdout(io.StringIO()):
        return hp.MFEngine(cfg)


def _state(eng):
    eng.flush_lazy()
    opt = eng.optimizer
    return (eng.model.flat.clone(), None if opt.exp_avg is None else opt.exp_avg.clone(),
            None if opt.exp_avg_sq is None else opt.exp_avg_sq.clone())


@pytest.mark.parametrize("optimizer,extra", [("adam", {}), ("adam", {"dense_opt": "lazy"}), ("rmsprop", {"dense_opt": "lazy"}),
                                             ("sgd", {"sgd_mode": "owned"})])
def test_resume_continues_the_run_bit_for_bit(hip_device, tmp_path, optimizer, extra):
    import beta_recsys_amd as hp

    U, I, D, B, steps = 900, 700, 64, 128, 3
    rng = np.random.default_rng(5)
    w0 = onp.init_params(U, I, D, seed=4)
    epochs = [tuple(torch.from_numpy(a).cuda() for a in _unique_row_batches(rng, U, I, B, steps)) for _ in range(2)]

    def train(eng, triples, e):
        with contextlib.redirect_stdout(io.StringIO()):
            eng.train_an_epoch(hp.DeviceTripleBatcher(*triples, B, shuffle=False), e)

    ref = _engine(U, I, D, B, optimizer, **extra)
    ref.model.load_state_dict({k: torch.from_numpy(v) for k, v in w0.items()})
    train(ref, epochs[0], 0)
    train(ref, epochs[1], 1)
    want = _state(ref)
    assert (ref._lazy is not None) == ("dense_opt" in extra)

    a = _engine(U, I, D, B, optimizer, **extra)
    a.model.load_state_dict({k: torch.from_numpy(v) for k, v in w0.items()})
    train(a, epochs[0], 0)
    path = str(tmp_path / "mf.ckpt")
    a.save_checkpoint(path, optimizer_state=True)
    # the first file is the reference's format: a plain state_dict with its five keys
    sd = torch.load(path, map_location="cpu")
    assert sorted(sd) == sorted(a.model.state_dict()) and os.path.exists(path + ".opt")
    del a
    b = _engine(U, I, D, B, optimizer, **extra)
    b.resume_checkpoint(path, optimizer_state=True)
    assert b.epoch_stats().step == steps
    train(b, epochs[1], 1)
    got = _state(b)
    for name, x, y in zip(("weights", "exp_avg", "exp_avg_sq"), got, want):
        assert (x is None) == (y is None)
        if x is not None:
            assert torch.equal(x, y), f"{name}: {int((x != y).sum())} elements differ after the resumed epoch"
    # without the second file the moments start from zero: the run is a different one (what the reference's own
    # checkpoint gives), so the comparison above is not vacuous
    if optimizer != "sgd":
        c = _engine(U, I, D, B, optimizer, **extra)
        c.resume_checkpoint(path)
        train(c, epochs[1], 1)
        assert not torch.equal(_state(c)[0], want[0])


@pytest.mark.parametrize("optimizer_state", [True, False])
def test_resume_into_an_engine_with_pending_lazy_replays(hip_device, tmp_path, optimizer_state):
    """resume_checkpoint used to replace the weights FIRST and flush afterwards, so an engine whose lazy
    state was dirty (an epoch run in pieces) replayed its OLD moments onto the just-restored weights.  The restored
    weights must be the file's, bit for bit, and a later flush must leave them alone."""
    import beta_recsys_amd as hp

    U, I, D, B, steps = 900, 700, 64, 128, 4
    rng = np.random.default_rng(11)
    w0 = onp.init_params(U, I, D, seed=8)
    triples = tuple(torch.from_numpy(a).cuda() for a in _unique_row_batches(rng, U, I, B, steps))
    src = _engine(U, I, D, B, "adam", dense_opt="lazy")
    src.model.load_state_dict({k: torch.from_numpy(v) for k, v in w0.items()})
    with contextlib.redirect_stdout(io.StringIO()):
        src.train_an_epoch(hp.DeviceTripleBatcher(*triples, B, shuffle=False), 0)
    path = str(tmp_path / "mf.ckpt")
    src.save_checkpoint(path, optimizer_state=True)
    want = {k: v.clone() for k, v in src.model.state_dict().items()}

    dirty = _engine(U, I, D, B, "adam", dense_opt="lazy")
    other = tuple(torch.from_numpy(a).cuda() for a in _unique_row_batches(np.random.default_rng(12), U, I, B, steps))
    with contextlib.redirect_stdout(io.StringIO()):
        prepared = dirty.prepare_epoch(hp.DeviceTripleBatcher(*other, B, shuffle=False))
        dirty.run_prepared_epoch(prepared, sync=False, steps=(0, steps - 1))     # the epoch's flush has not run
    assert dirty._lazy["dirty"]
    dirty.resume_checkpoint(path, optimizer_state=optimizer_state)
    assert not dirty._lazy["dirty"]
    dirty.flush_lazy()
    dirty._lazy["dirty"] = True      # force the flush launch itself: nothing may be owed to any row
    dirty.flush_lazy()
    torch.cuda.synchronize()
    for k, v in dirty.model.state_dict().items():
        assert torch.equal(v, want[k]), f"{k}: {int((v != want[k]).sum())} restored elements were perturbed"


def test_lazy_adam_resumes_beyond_the_scalars_table(hip_device):
    """a restored clock beyond the 65 536 tabulated steps (or a dense sweep at step 65 535) left the table's
    last entry zero and every later lazy step raised HIPREC_STATUS_LAZY_TABLE.  The table is created with its converged
    tail, and betas that do not converge inside it are refused when the state is set up."""
    import beta_recsys_amd as hp

    U, I, D, B = 400, 300, 64, 64
    rng = np.random.default_rng(1)
    eng = _engine(U, I, D, B, "adam", dense_opt="lazy")
    eng.load_optimizer_state(70_000)
    triples = tuple(torch.from_numpy(a).cuda() for a in _unique_row_batches(rng, U, I, B, 4))
    with contextlib.redirect_stdout(io.StringIO()):
        eng.train_an_epoch(hp.DeviceTripleBatcher(*triples, B, shuffle=False), 0)     # raised RuntimeError before
    st = eng.epoch_stats()
    assert st.step == 70_004 and np.isfinite(st.loss)
    bad = _engine(U, I, D, B, "adam", dense_opt="lazy")
    bad.optimizer.beta2 = 0.99999
    with pytest.raises(ValueError, match="converged"):
        bad._setup()
    auto = _engine(U, I, D, B, "adam")           # "auto" falls back to the sweep instead (tables this small sweep anyway)
    auto.optimizer.beta2 = 0.99999
    auto._setup()
    assert auto._lazy is None


@pytest.fixture(scope="module")
def nccl_group(hip_device):
    import socket

    import torch.distributed as dist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=hip_device)
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("optimizer,dense_opt", [("adam", "lazy"), ("adam", "sweep"), ("sgd", "auto")])
def test_sharded_engine_resumes_bit_for_bit(nccl_group, hip_device, tmp_path, optimizer, dense_opt):
    """The row-sharded engine: one reference-format file from the gathered shards + one optimizer file per rank."""
    import beta_recsys_amd as hp
    from beta_recsys_amd.sharded import ShardedMFEngine

    U, I, D, B, steps = 900, 700, 64, 128, 3
    rng = np.random.default_rng(9)
    w0 = {k: torch.from_numpy(v) for k, v in onp.init_params(U, I, D, seed=6).items()}
    epochs = [tuple(torch.from_numpy(a).cuda() for a in _unique_row_batches(rng, U, I, B, steps)) for _ in range(2)]
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer=optimizer, lr=0.02, batch_size=B,
                         loss="bpr", dense_opt=dense_opt, sgd_mode="rows"), "system": {"run_dir": "/tmp/hiprec_test_runs"}}

    def make():
        with contextlib.redirect_stdout(io.StringIO()):
            return ShardedMFEngine(cfg, full_state=w0)

    def train(eng, triples, e):
        with contextlib.redirect_stdout(io.StringIO()):
            eng.train_an_epoch(hp.DeviceTripleBatcher(*triples, B, shuffle=False), e)

    def state(eng):
        full = eng.gather_full_state_dict()
        opt = eng.optimizer
        return [full[k].clone() for k in sorted(full)] + [t.clone() for t in (opt.exp_avg, opt.exp_avg_sq) if t is not None]

    ref = make()
    train(ref, epochs[0], 0)
    train(ref, epochs[1], 1)
    want = state(ref)
    a = make()
    train(a, epochs[0], 0)
    path = str(tmp_path / "sharded.ckpt")
    a.save_checkpoint(path, optimizer_state=True)
    assert sorted(torch.load(path, map_location="cpu")) == sorted(w0) and os.path.exists(path + ".opt.rank0of1")
    del a
    with contextlib.redirect_stdout(io.StringIO()):
        b = ShardedMFEngine(cfg)
    b.resume_checkpoint(path, optimizer_state=True)
    train(b, epochs[1], 1)
    for x, y in zip(state(b), want):
        assert torch.equal(x, y), f"{int((x != y).sum())} elements differ after the resumed epoch"
