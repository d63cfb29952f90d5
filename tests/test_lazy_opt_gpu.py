"""Exact lazy Adam / RMSprop (csrc/lazy_opt.hip) against the dense sweep it replaces.

Reference semantics: beta_rec/models/torch_engine.py:30-39 -- torch.optim.Adam / RMSprop on DENSE nn.Embedding
gradients step every element every step.  The lazy form postpones a row's zero-gradient steps and replays them
(the same fp32 operations, the same per-step bias corrections) when the row is needed: after a flush the tables
and the moments must equal the dense sweeps' BIT FOR BIT, and a row that a step reads must already hold the bits
the sweeps would have given it."""
import contextlib
import ctypes
import io
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import REL  # noqa: F401  (north_star: 1e-5 relative)
from helpers import KEYS, assert_on_trajectory, assert_scalar_close, mf_trajectory
from oracle import mf_numpy as onp

pytestmark = pytest.mark.gpu
KIND = {"adam": 1, "rmsprop": 2}


def new_stats(lib, _lib, dev):
    from beta_recsys_amd.mf import _new_stats

    stats = _new_stats(dev)
    _lib.check(lib.hiprec_stats_reset(_lib.ptr(stats), 0.9, 0.999, _lib.stream_ptr(dev)))
    return stats


class Lazy:
    """hiprec_lazy_state over flat test buffers."""

    def __init__(self, lib, _lib, w, g, m, v, U, I, D, kind, lr, cap=64):
        self.lib, self._lib, self.dev = lib, _lib, w.device
        self.stamp_u = torch.full((U,), -1, dtype=torch.int32, device=w.device)
        self.stamp_i = torch.full((I,), -1, dtype=torch.int32, device=w.device)
        self.scalars = torch.zeros((cap, 2), dtype=torch.float32, device=w.device)
        self.c = _lib.LazyState(w.data_ptr(), g.data_ptr(), m.data_ptr() if m is not None else None, v.data_ptr(), U, I, D,
                                kind, self.stamp_u.data_ptr(), self.stamp_i.data_ptr(), self.scalars.data_ptr(), cap, 0, lr,
                                0.9, 0.999, 1e-8)

    def rows(self, users, items_a, items_b, items_c):
        self._keep = (users, items_a, items_b, items_c)
        p = lambda t: t.data_ptr() if t is not None and t.numel() else None  # noqa: E731
        n = lambda t: t.numel() if t is not None else 0  # noqa: E731
        return self._lib.LazyRows(p(users), n(users), p(items_a), n(items_a), p(items_b), n(items_b), p(items_c),
                                  n(items_c))

    def catchup(self, stats, *lists):
        r = self.rows(*lists)
        self._lib.check(self.lib.hiprec_lazy_catchup(ctypes.byref(self.c), ctypes.byref(r), self._lib.ptr(stats),
                                                     self._lib.stream_ptr(self.dev)))

    def update(self, stats, *lists):
        r = self.rows(*lists)
        self._lib.check(self.lib.hiprec_lazy_update(ctypes.byref(self.c), ctypes.byref(r), None, self._lib.ptr(stats),
                                                    self._lib.stream_ptr(self.dev)))

    def flush(self, stats):
        self._lib.check(self.lib.hiprec_lazy_flush(ctypes.byref(self.c), self._lib.ptr(stats),
                                                   self._lib.stream_ptr(self.dev)))


@pytest.mark.parametrize("D", [8, 10, 64, 100, 128, 254, 256])
@pytest.mark.parametrize("opt", ["adam", "rmsprop"])
def test_lazy_rows_equal_the_dense_sweeps_bit_for_bit(hip_device, opt, D):
    """40 steps on a 50 x 30 table, each touching a few rows: some rows every step, some twice with a gap of 30+
    steps, some never; the lists carry duplicates and -1 padding and use all four list kinds; the table of per-step
    scalars is SHORTER than the run is long only where that is legal (not here: cap 64).  In lockstep with dense
    sweeps fed the same gradients: (1) after every catch-up the step's rows hold the dense sweep's bits (what the
    gradient kernel would read), (2) after the flush w, m, v are bit-identical everywhere and g is all zero, (3)
    never-touched rows keep their stamp -1 and their bits."""
    from beta_recsys_amd import _lib

    lib, dev = _lib.load(), hip_device
    U, I, T, lr = 50, 30, 40, 0.05
    kind = KIND[opt]
    P = (U + I) * (D + 1) + 1
    gen = torch.Generator(device="cuda").manual_seed(D + kind)
    w0 = torch.randn(P, device=dev, generator=gen) * 0.1
    wd, md, vd, gd = w0.clone(), torch.zeros_like(w0), torch.zeros_like(w0), torch.zeros_like(w0)
    wl, ml, vl, gl = w0.clone(), torch.zeros_like(w0), torch.zeros_like(w0), torch.zeros_like(w0)
    sd, sl = new_stats(lib, _lib, dev), new_stats(lib, _lib, dev)
    lazy = Lazy(lib, _lib, wl, gl, ml if opt == "adam" else None, vl, U, I, D, kind, lr)
    st = _lib.stream_ptr(dev)
    rng = np.random.default_rng(D)

    def flat_index(users, items):
        rows = []
        for u in users:
            rows += list(range(u * D, (u + 1) * D)) + [(U + I) * D + u]
        for i in items:
            rows += list(range(U * D + i * D, U * D + (i + 1) * D)) + [(U + I) * D + U + i]
        return torch.tensor(sorted(set(rows)), dtype=torch.int64, device=dev)

    hot_u, hot_i = [0, 1], [0]
    for t in range(1, T + 1):
        users = set(hot_u) | set(rng.integers(2, 20, 2).tolist())
        items = set(hot_i) | set(rng.integers(1, 12, 3).tolist())
        if t in (3, 36):
            users |= {40, 41}
            items |= {25}          # touched twice, 33 steps apart
        if t == 39:
            users |= {45}          # first touch near the end
        users, items = sorted(users), sorted(items)
        # lists with duplicates and padding, spread over the four list kinds
        lu = torch.tensor(users + users[:2] + [-1], dtype=torch.int64, device=dev)
        la = torch.tensor(items[::2] + [-1, items[0]], dtype=torch.int64, device=dev)
        lb = torch.tensor(items[1::2], dtype=torch.int64, device=dev)
        lc = torch.tensor(items[:1] + [-1] + items[-1:], dtype=torch.int32, device=dev)
        idx = flat_index(users, items)
        lazy.catchup(sl, lu, la, lb, lc)
        # what the step's gradient kernel reads: w (the moments are replayed by the step's update)
        assert torch.equal(wl[idx], wd[idx]), f"step {t}: a caught-up row's weights differ from the dense sweeps'"
        g = torch.zeros(P, device=dev)
        g[idx] = torch.randn(idx.numel(), device=dev, generator=gen) * 0.01
        g[-1] = float(rng.normal()) * 0.01
        gd.copy_(g)
        gl.copy_(g)
        for stats in (sd, sl):
            _lib.check(lib.hiprec_stats_advance_step(_lib.ptr(stats), st))
        _lib.check(lib.hiprec_opt_dense_step(kind, _lib.ptr(wd), _lib.ptr(gd), _lib.ptr(md), _lib.ptr(vd), P, lr, 0.9, 0.999,
                                             1e-8, _lib.ptr(sd), None, -1, st))
        lazy.update(sl, lu, la, lb, lc)
        assert torch.equal(wl[idx], wd[idx]) and torch.equal(vl[idx], vd[idx]) and torch.equal(ml[idx], md[idx]), \
            f"step {t}: an updated row differs"
        assert float(gl.abs().max()) == 0.0
    # untouched rows lag behind until the flush
    lag = flat_index([40], [25])
    assert not torch.equal(vl[lag], vd[lag])
    lazy.flush(sl)
    assert torch.equal(wl, wd), f"w differs in {int((wl != wd).sum())} elements after the flush"
    assert torch.equal(vl, vd) and (opt != "adam" or torch.equal(ml, md))
    never_u, never_i = [48, 49], [28, 29]
    assert all(int(lazy.stamp_u[u]) == -1 for u in never_u) and all(int(lazy.stamp_i[i]) == -1 for i in never_i)
    assert int(lazy.stamp_u[40]) == T and int(lazy.stamp_i[25]) == T and int(lazy.stamp_u[0]) == T
    assert torch.equal(wl[flat_index(never_u, never_i)], w0[flat_index(never_u, never_i)])
    # a second flush has nothing to do
    before = wl.clone()
    lazy.flush(sl)
    assert torch.equal(wl, before)


@pytest.mark.parametrize("opt,D", [("adam", 128), ("adam", 64), ("rmsprop", 128), ("adam", 256)])
def test_lazy_rows_in_many_chunks_with_long_gaps_bit_for_bit(hip_device, opt, D):
    """The same lockstep comparison where the vector kernel's machinery is all in play: 3 000 x 500 rows, 150 steps of
    ~260 list entries each (several 64-entry chunks per launch, their rows ranked by stamp and dealt out to the waves),
    gaps beyond 64 steps (the replay fetches its scalars in blocks of 64 steps), and -- after a flush + mark_current at
    step 20, which stamps never-touched rows (all moments zero) as current -- groups whose OLDEST row has nothing to
    replay (the walk then starts later than the scalars fetched with the rows).  After every catch-up the step's rows
    hold the dense sweeps' weights; after the final flush w, m, v are bit-identical everywhere."""
    from beta_recsys_amd import _lib

    lib, dev = _lib.load(), hip_device
    U, I, T, lr = 3000, 500, 150, 0.05
    kind = KIND[opt]
    P = (U + I) * (D + 1) + 1
    gen = torch.Generator(device="cuda").manual_seed(7 * D + kind)
    w0 = torch.randn(P, device=dev, generator=gen) * 0.1
    wd, md, vd, gd = w0.clone(), torch.zeros_like(w0), torch.zeros_like(w0), torch.zeros_like(w0)
    wl, ml, vl, gl = w0.clone(), torch.zeros_like(w0), torch.zeros_like(w0), torch.zeros_like(w0)
    sd, sl = new_stats(lib, _lib, dev), new_stats(lib, _lib, dev)
    lazy = Lazy(lib, _lib, wl, gl, ml if opt == "adam" else None, vl, U, I, D, kind, lr, cap=256)
    st = _lib.stream_ptr(dev)
    rng = np.random.default_rng(D + kind)
    cols = torch.arange(D, device=dev)

    def flat_index(users, items):
        u = torch.as_tensor(users, dtype=torch.int64, device=dev)
        i = torch.as_tensor(items, dtype=torch.int64, device=dev)
        parts = [(u[:, None] * D + cols).ravel(), (U * D + i[:, None] * D + cols).ravel(), (U + I) * D + u,
                 (U + I) * D + U + i]
        return torch.unique(torch.cat(parts))

    for t in range(1, T + 1):
        # users: a hot set every step, a Zipf-ish draw, and a slice that comes back only every ~70 steps
        users = set(range(5)) | set(rng.integers(5, 600, 150).tolist()) | set((600 + (t % 70) * 30 + np.arange(30)).tolist())
        items = set(range(3)) | set(rng.integers(3, 400, 80).tolist())
        if t > 20:
            users |= set(rng.integers(2800, 3000, 6).tolist())   # stamped current at step 20 with zero moments
        users, items = sorted(users), sorted(items)
        lu = torch.tensor(users + users[:7] + [-1], dtype=torch.int64, device=dev)
        la = torch.tensor(items[::2] + [-1, items[0]], dtype=torch.int64, device=dev)
        lb = torch.tensor(items[1::2], dtype=torch.int64, device=dev)
        lc = torch.tensor(items[:9] + [-1] + items[-3:], dtype=torch.int32, device=dev)
        idx = flat_index(users, items)
        lazy.catchup(sl, lu, la, lb, lc)
        assert torch.equal(wl[idx], wd[idx]), f"step {t}: a caught-up row's weights differ from the dense sweeps'"
        g = torch.zeros(P, device=dev)
        g[idx] = torch.randn(idx.numel(), device=dev, generator=gen) * 0.01
        g[-1] = float(rng.normal()) * 0.01
        gd.copy_(g)
        gl.copy_(g)
        for stats in (sd, sl):
            _lib.check(lib.hiprec_stats_advance_step(_lib.ptr(stats), st))
        _lib.check(lib.hiprec_opt_dense_step(kind, _lib.ptr(wd), _lib.ptr(gd), _lib.ptr(md), _lib.ptr(vd), P, lr, 0.9, 0.999,
                                             1e-8, _lib.ptr(sd), None, -1, st))
        lazy.update(sl, lu, la, lb, lc)
        assert torch.equal(wl[idx], wd[idx]) and torch.equal(vl[idx], vd[idx]) and torch.equal(ml[idx], md[idx]), \
            f"step {t}: an updated row differs"
        if t == 20:   # what a dense step in between leaves: everything flushed, EVERY row stamped current
            lazy.flush(sl)
            assert torch.equal(wl, wd) and torch.equal(vl, vd)
            _lib.check(lib.hiprec_lazy_mark_current(ctypes.byref(lazy.c), _lib.ptr(sl), st))
            assert int(lazy.stamp_u[2999]) == 20
    assert float(gl.abs().max()) == 0.0
    lazy.flush(sl)
    assert torch.equal(wl, wd), f"w differs in {int((wl != wd).sum())} elements after the flush"
    assert torch.equal(vl, vd) and (opt != "adam" or torch.equal(ml, md))


@pytest.mark.parametrize("opt,D", [("adam", 128), ("adam", 10), ("adam", 64), ("rmsprop", 64)])
def test_bounded_replay_of_rows_that_lag_for_hundreds_of_steps_bit_for_bit(hip_device, opt, D):
    """BOUNDED REPLAY (csrc/lazy_opt.hip): a row's zero-gradient steps are replayed in full only until its weights stop
    moving, after that only the moments decay -- which must still be the dense sweeps' bits.  700 steps in lockstep with
    dense sweeps; blocks of rows are touched once early (steps 3-8, 30-33) and again after 250-690 steps, others get
    their first gradient late; among the weights are zeros, denormals and values around 1e-31 (below the magnitude
    from which the early exit may be taken), and gradients from 1e-12 to 10 (moments that underflow, and moments that
    keep the weights moving for longer than the typical 150-200 steps; the exit is not looked for before a walk is 192
    steps long).  After every catch-up the step's rows hold the
    dense sweeps' weights; after the final flush w, m, v are bit-identical everywhere."""
    from beta_recsys_amd import _lib

    lib, dev = _lib.load(), hip_device
    U, I, T, lr = 260, 90, 700, 0.05
    kind = KIND[opt]
    P = (U + I) * (D + 1) + 1
    gen = torch.Generator(device="cuda").manual_seed(11 * D + kind)
    w0 = torch.randn(P, device=dev, generator=gen) * 0.1
    w0[: 40 * D] = 0.0                                         # users 0-39: zero weights
    w0[40 * D: 60 * D] *= 1e-30                                # users 40-59: around 1e-31
    w0[60 * D: 70 * D] = 1e-42                                 # users 60-69: denormal
    wd, md, vd, gd = w0.clone(), torch.zeros_like(w0), torch.zeros_like(w0), torch.zeros_like(w0)
    wl, ml, vl, gl = w0.clone(), torch.zeros_like(w0), torch.zeros_like(w0), torch.zeros_like(w0)
    sd, sl = new_stats(lib, _lib, dev), new_stats(lib, _lib, dev)
    lazy = Lazy(lib, _lib, wl, gl, ml if opt == "adam" else None, vl, U, I, D, kind, lr, cap=1024)
    st = _lib.stream_ptr(dev)
    rng = np.random.default_rng(D + kind)
    cols = torch.arange(D, device=dev)

    def flat_index(users, items):
        u = torch.as_tensor(users, dtype=torch.int64, device=dev)
        i = torch.as_tensor(items, dtype=torch.int64, device=dev)
        return torch.unique(torch.cat([(u[:, None] * D + cols).ravel(), (U * D + i[:, None] * D + cols).ravel(),
                                       (U + I) * D + u, (U + I) * D + U + i]))

    early = {3: range(0, 35), 4: range(35, 70), 5: range(70, 100), 8: range(100, 130), 30: range(130, 160),
             33: range(160, 190)}
    late = {260: range(0, 20), 400: range(20, 60), 520: range(60, 110), 640: range(110, 150), 690: range(150, 200)}
    for t in range(1, T + 1):
        users = set(range(250, 256)) | set(rng.integers(200, 250, 4).tolist())      # a hot set and a warm one
        items = set(range(4)) | set(rng.integers(4, 40, 6).tolist())
        for table in (early, late):
            if t in table:
                users |= set(table[t])
                items |= {40 + (j % 50) for j in table[t]}
        users, items = sorted(users), sorted(items)
        lu = torch.tensor(users + users[:3] + [-1], dtype=torch.int64, device=dev)
        la = torch.tensor(items[::2] + [-1, items[0]], dtype=torch.int64, device=dev)
        lb = torch.tensor(items[1::2], dtype=torch.int64, device=dev)
        lc = torch.tensor(items[:2] + [-1], dtype=torch.int32, device=dev)
        idx = flat_index(users, items)
        lazy.catchup(sl, lu, la, lb, lc)
        assert torch.equal(wl[idx], wd[idx]), f"step {t}: a caught-up row's weights differ from the dense sweeps'"
        g = torch.zeros(P, device=dev)
        scale = 10.0 ** float(rng.integers(-12, 2)) if t in early or t in late else 0.01
        g[idx] = torch.randn(idx.numel(), device=dev, generator=gen) * scale
        g[-1] = float(rng.normal()) * 0.01
        gd.copy_(g)
        gl.copy_(g)
        for stats in (sd, sl):
            _lib.check(lib.hiprec_stats_advance_step(_lib.ptr(stats), st))
        _lib.check(lib.hiprec_opt_dense_step(kind, _lib.ptr(wd), _lib.ptr(gd), _lib.ptr(md), _lib.ptr(vd), P, lr, 0.9, 0.999,
                                             1e-8, _lib.ptr(sd), None, -1, st))
        lazy.update(sl, lu, la, lb, lc)
        if t in early or t in late or t % 50 == 0:
            assert torch.equal(wl[idx], wd[idx]) and torch.equal(vl[idx], vd[idx]) and torch.equal(ml[idx], md[idx]), \
                f"step {t}: an updated row differs"
    lazy.flush(sl)
    assert torch.equal(wl, wd), f"w differs in {int((wl != wd).sum())} elements after the flush"
    assert torch.equal(vl, vd) and (opt != "adam" or torch.equal(ml, md))
    assert float(gl.abs().max()) == 0.0


def test_lazy_adam_beyond_the_scalars_table(hip_device):
    """A table of 8 entries and 12 steps: bias corrections still move at step 8, so the update kernel must raise
    HIPREC_STATUS_LAZY_TABLE instead of letting a later replay use the wrong scalars."""
    from beta_recsys_amd import _lib
    from beta_recsys_amd.mf import read_stats

    lib, dev = _lib.load(), hip_device
    U, I, D = 4, 4, 8
    P = (U + I) * (D + 1) + 1
    w, g, m, v = (torch.zeros(P, device=dev) for _ in range(4))
    stats = new_stats(lib, _lib, dev)
    lazy = Lazy(lib, _lib, w, g, m, v, U, I, D, 1, 0.05, cap=8)
    lu = torch.tensor([0], dtype=torch.int64, device=dev)
    for t in range(12):
        _lib.check(lib.hiprec_stats_advance_step(_lib.ptr(stats), _lib.stream_ptr(dev)))
        lazy.update(stats, lu, None, None, None)
        assert bool(read_stats(stats).status & _lib.STATUS_LAZY_TABLE) == (t + 1 >= 8)


def planned_epochs(hip_device, optimizer, lr, dense_opt, epochs=2, D=64, driver="c"):
    import beta_recsys_amd as hp
    from beta_recsys_amd.sharded import ShardedMFEngine

    U, I, B = 3000, 400, 256
    n = 4 * B + B // 3
    w0 = onp.init_params(U, I, D, seed=3)
    rng = np.random.default_rng(D)
    p = 1.0 / np.arange(1, I + 1)
    users, pos, neg = rng.integers(0, U // 3, n) * 3 % U, rng.choice(I, n, p=p / p.sum()), rng.integers(0, I, n)
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer=optimizer, lr=lr, batch_size=B,
                         loss="bpr", step_driver=driver, dense_opt=dense_opt), "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ShardedMFEngine(cfg, full_state={k: torch.from_numpy(v) for k, v in w0.items()})
    assert (eng._lazy is not None) == (dense_opt == "lazy")
    loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a).cuda() for a in (users, pos, neg)), B, shuffle=False)
    sums = []
    with contextlib.redirect_stdout(io.StringIO()):
        for e in range(epochs):
            sums.append(eng.train_an_epoch(loader, e))
    batches = [(users[k:k + B], pos[k:k + B], neg[k:k + B]) for _ in range(epochs) for k in range(0, n, B)]
    full = {k: v.cpu().numpy() for k, v in eng.gather_full_state_dict().items()}
    state = (eng.model.flat.clone(), eng.optimizer.exp_avg_sq.clone(),
             None if eng.optimizer.exp_avg is None else eng.optimizer.exp_avg.clone())
    return eng, w0, batches, sums, full, state


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


@pytest.mark.parametrize("driver", ["c", "torch"])
@pytest.mark.parametrize("optimizer,lr", [("adam", 0.05), ("rmsprop", 0.01)])
def test_planned_epochs_with_lazy_state_follow_the_oracle_and_the_sweep(nccl_group, hip_device, optimizer, lr, driver):
    """ShardedMFEngine with ``dense_opt: "lazy"`` (two planned epochs, a third of the users never drawn, Zipf items):
    the loss sums and the gathered state_dict lie on the oracle's trajectory like the sweep's; against the same engine
    with ``dense_opt: "sweep"`` the never-touched rows are bit-identical and the rest agrees to the accuracy atomics
    leave (the two runs add their gradients in different orders)."""
    eng, w0, batches, sums, full, state = planned_epochs(hip_device, optimizer, lr, "lazy", driver=driver)
    assert eng._step_mode == driver
    per_epoch = len(batches) // 2
    w = onp.copy_params(w0)
    st = onp.new_opt_state(w, optimizer)
    for e in range(2):
        tot_loss = tot_reg = 0.0
        for batch in batches[e * per_epoch:(e + 1) * per_epoch]:
            loss, reg = onp.mf_train_step(w, st, batch, "bpr", optimizer, lr)
            tot_loss += loss
            tot_reg += reg
        assert_scalar_close(sums[e][0], tot_loss, REL, f"epoch {e} loss sum")
        assert_scalar_close(sums[e][1], tot_reg, REL, f"epoch {e} regularizer sum")
    w_ref, env, upd = mf_trajectory(w0, batches, optimizer, lr)
    assert_on_trajectory(full, w_ref, env, upd, f"lazy {optimizer}, {driver} driver")
    assert float(eng._g_flat.abs().max()) == 0.0, "every consumed gradient row is cleared"
    assert not eng._lazy["dirty"], "the epoch ends with a flush"
    # users = 3k only: two thirds of the user rows are never touched -> stamp -1, weights bit-identical to the start
    su = eng._lazy["stamp_u"].cpu().numpy()
    assert (su[1::3] == -1).all() and (su[2::3] == -1).all() and (su[::3] >= -1).all()
    assert np.array_equal(full["user_emb.weight"][1::3], w0["user_emb.weight"][1::3])
    _, _, _, sums_s, full_s, _ = planned_epochs(hip_device, optimizer, lr, "sweep", driver=driver)
    for e in range(2):
        assert_scalar_close(sums[e][0], sums_s[e][0], 1e-5, "lazy vs sweep epoch loss")
    assert_on_trajectory(full_s, w_ref, env, upd, f"sweep {optimizer}")
    for k in KEYS:
        assert np.array_equal(full[k] == w0[k], full_s[k] == w0[k]), f"{k}: lazy and sweep moved different elements"


def test_lazy_state_survives_a_dense_step_in_between(nccl_group, hip_device):
    """A per-batch step (dense sweep) between two lazy planned epochs: the engine flushes before the sweep and marks
    every touched row current after it; the trajectory stays the oracle's."""
    eng, w0, batches, _, _, _ = planned_epochs(hip_device, "adam", 0.05, "lazy", epochs=1)
    rng = np.random.default_rng(5)
    extra = (rng.integers(0, 3000, 256), rng.integers(0, 400, 256), rng.integers(0, 400, 256))
    eng.train_single_batch(tuple(torch.from_numpy(a) for a in extra))
    clock = int(eng.step_count)
    su = eng._lazy["stamp_u"].cpu().numpy()
    assert set(np.unique(su)) == {clock}, "after a dense sweep every row is current as of the clock"
    import beta_recsys_amd as hp

    users, pos, neg = (np.concatenate([b[j] for b in batches]) for j in range(3))
    loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a).cuda() for a in (users, pos, neg)), 256, shuffle=False)
    with contextlib.redirect_stdout(io.StringIO()):
        eng.train_an_epoch(loader, 1)
    full = {k: v.cpu().numpy() for k, v in eng.gather_full_state_dict().items()}
    w_ref, env, upd = mf_trajectory(w0, batches + [extra] + batches, "adam", 0.05)
    assert_on_trajectory(full, w_ref, env, upd, "lazy epochs around a dense step")


def test_lazy_parity_against_the_ieee_arithmetic_build(hip_device):
    """The bit-for-bit test again with libhiprec_ieee.so (ATen's correctly rounded sqrt / division): the replay uses
    whatever arithmetic the sweep uses."""
    from beta_recsys_amd import _lib

    if os.environ.get("HIPREC_LIB", "").endswith("ieee.so"):
        pytest.skip("already running against the IEEE build")
    ieee = os.path.join(os.path.dirname(_lib.LIB_PATH), "libhiprec_ieee.so")
    assert os.path.exists(ieee), "libhiprec_ieee.so is missing: run __graft_entry__.build()"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k",
                          "bit_for_bit"], env=dict(os.environ, HIPREC_LIB="libhiprec_ieee.so"), capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


# ---- MFEngine (the un-sharded drop-in engine) with the lazy optimizer ------------------------------------------------

def mf_lazy_run(optimizer, lr, loss, dense_opt, pieces=None, single_between=False, reg=None, lazy_grad="owned"):
    import beta_recsys_amd as hp
    from test_mf_gpu import get_weights, load_weights, make_engine

    U, I, D, B = 3000, 400, 64, 256
    n = 5 * B + 77
    w0 = onp.init_params(U, I, D, seed=4)
    rng = np.random.default_rng(1)
    p = 1.0 / np.arange(1, I + 1)
    users, pos = rng.integers(0, U // 2, n) * 2, rng.choice(I, n, p=p / p.sum())      # odd users are never drawn
    third = rng.integers(0, I, n) if loss == "bpr" else (rng.random(n) < 0.3).astype(np.float32)
    eng = make_engine(U, I, D, optimizer, loss, lr, B, reg=reg, dense_opt=dense_opt, prefetch_epoch=False,
                      lazy_grad=lazy_grad)
    load_weights(eng, w0)
    eng._setup()
    assert (eng._lazy is not None) == (dense_opt == "lazy")
    assert eng._lazy_owned() == (lazy_grad if dense_opt == "lazy" and loss == "bpr" and lazy_grad != "atomic" else False)
    if loss == "bpr":
        loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a).cuda() for a in (users, pos, third)), B, shuffle=False)
    else:
        from beta_recsys_amd.data import DeviceTensorBatcher

        loader = DeviceTensorBatcher((torch.from_numpy(users).cuda(), torch.from_numpy(pos).cuda(),
                                      torch.from_numpy(third).cuda()), B, shuffle=False)
    batches = [(users[k:k + B], pos[k:k + B], third[k:k + B]) for k in range(0, n, B)]
    visited, sums = [], []
    with contextlib.redirect_stdout(io.StringIO()):
        for epoch in range(2):
            if pieces and epoch == 1:
                prepared = eng.prepare_epoch(loader)
                for piece in pieces:
                    eng.run_prepared_epoch(prepared, sync=False, steps=piece)
                sums.append(eng.epoch_stats().loss_sum)
            else:
                eng.train_an_epoch(loader, epoch)
                sums.append(eng.writer.scalars[-2][1])
            visited += batches
            if single_between and epoch == 0:
                extra = (rng.integers(0, U, B), rng.integers(0, I, B),
                         rng.integers(0, I, B) if loss == "bpr" else (rng.random(B) < 0.3).astype(np.float32))
                eng.train_single_batch(tuple(torch.from_numpy(a) for a in extra))
                visited.append(extra)
    return eng, w0, visited, sums, get_weights(eng)


@pytest.mark.parametrize("optimizer,lr,loss,lazy_grad,reg", [
    ("adam", 0.05, "bpr", "pull", None), ("rmsprop", 0.01, "bpr", "pull", None), ("adam", 0.05, "bpr", "pull", 0.02),
    ("adam", 0.05, "bpr", "owned", None), ("adam", 0.05, "bpr", "atomic", None), ("rmsprop", 0.01, "bpr", "owned", None),
    ("rmsprop", 0.01, "bpr", "atomic", None), ("adam", 0.02, "bce", "owned", None), ("adam", 0.05, "bpr", "owned", 0.02)])
def test_mf_engine_epochs_with_the_lazy_optimizer(hip_device, optimizer, lr, loss, lazy_grad, reg):
    """MFEngine.train_an_epoch with ``dense_opt: "lazy"`` (hiprec_mf_epoch_lazy_pull: catch-up, gradient launch into the
    contribution buffer, ONE launch that sums / replays the moments / steps every row of the batch; or
    hiprec_mf_epoch_lazy / _lazy_owned: catch-up, gradient kernel -- the owned-rows kernel or mf_bpr_grad_kernel's
    atomics --, update per step; flush at the end of the epoch): epoch
    sums to 1e-5 of the oracle's, every weight on the oracle's trajectory, never-drawn users bit-identical with stamp
    -1, the gradient buffer clean -- like the dense-sweep engine."""
    eng, w0, visited, sums, got = mf_lazy_run(optimizer, lr, loss, "lazy", lazy_grad=lazy_grad, reg=reg)
    _, _, _, sums_s, got_s = mf_lazy_run(optimizer, lr, loss, "sweep", reg=reg)
    per = len(visited) // 2
    w = onp.copy_params(w0)
    st = onp.new_opt_state(w, optimizer)
    rc = reg or 0.0
    for e in range(2):
        tot = sum(onp.mf_train_step(w, st, b, loss, optimizer, lr, rc)[0] for b in visited[e * per:(e + 1) * per])
        assert_scalar_close(sums[e], tot, REL, f"epoch {e} loss sum (lazy)")
        assert_scalar_close(sums_s[e], tot, REL, f"epoch {e} loss sum (sweep)")
    traj = mf_trajectory(w0, visited, optimizer, lr, reg_coef=rc, loss=loss)
    assert_on_trajectory(got, *traj, f"lazy {optimizer} {loss}")
    assert_on_trajectory(got_s, *traj, f"sweep {optimizer} {loss}")
    assert float(eng._g_flat.abs().max()) == 0.0 and not eng._lazy["dirty"]
    su = eng._lazy["stamp_u"].cpu().numpy()
    assert (su[1::2] == -1).all() and np.array_equal(got["user_emb.weight"][1::2], w0["user_emb.weight"][1::2])


@pytest.mark.parametrize("optimizer,lr,D", [("adam", 0.05, 64), ("rmsprop", 0.01, 128), ("adam", 0.05, 256)])
def test_lazy_pull_step_is_the_three_launch_step_bit_for_bit_where_no_row_repeats(hip_device, optimizer, lr, D):
    """Batches in which no row occurs twice: every gradient element is ONE term, so the owner-pulls form (gradient parts
    through the contribution buffer + lazy_pull_apply_kernel) and the round-4 form (complete row gradients into the
    dense buffer + the update launch) must leave the same BITS -- weights, both moments, stamps -- including rows that
    lag several steps between two visits (their moments are replayed by the apply launch / the update launch)."""
    import beta_recsys_amd as hp
    from test_mf_gpu import load_weights, make_engine

    U, I, B, steps = 1500, 1200, 128, 4
    rng = np.random.default_rng(D)
    w0 = onp.init_params(U, I, D, seed=2)
    epochs = []
    for _ in range(3):
        us, ps, ns = [], [], []
        for _ in range(steps):
            us.append(rng.permutation(U // 2)[:B] * 2)          # few users: rows come back after a gap of a few steps
            items = rng.permutation(I)[: 2 * B]
            ps.append(items[:B])
            ns.append(items[B:])
        epochs.append(tuple(torch.from_numpy(np.concatenate(a).astype(np.int64)).cuda() for a in (us, ps, ns)))
    out = {}
    for form in ("pull", "owned"):
        eng = make_engine(U, I, D, optimizer, "bpr", lr, B, dense_opt="lazy", lazy_grad=form, prefetch_epoch=False)
        load_weights(eng, w0)
        eng._setup()
        assert eng._lazy_owned() == form
        with contextlib.redirect_stdout(io.StringIO()):
            for e, triples in enumerate(epochs):
                eng.train_an_epoch(hp.DeviceTripleBatcher(*triples, B, shuffle=False), e)
        opt = eng.optimizer
        out[form] = (eng.model.flat.clone(), opt.exp_avg_sq.clone(), None if opt.exp_avg is None else opt.exp_avg.clone(),
                     eng._lazy["stamp_u"].clone(), eng._lazy["stamp_i"].clone(), eng.epoch_stats().loss_sum)
        assert float(eng._g_flat.abs().max()) == 0.0
    for name, a, b in zip(("w", "v", "m", "stamp_u", "stamp_i"), out["pull"], out["owned"]):
        if a is not None:
            assert torch.equal(a, b), f"{name}: {int((a != b).sum())} elements differ between the two forms"
    assert out["pull"][5] == pytest.approx(out["owned"][5], rel=1e-6)


def test_mf_engine_lazy_epoch_in_pieces_and_around_a_dense_step(hip_device):
    """The second epoch enqueued in pieces (the flush rides on the piece that reaches the last step) and a per-batch
    step (dense sweep: flush before, every row marked current after) between the epochs; optimizer_state() reads
    flushed moments."""
    eng, w0, visited, _, got = mf_lazy_run("adam", 0.05, "bpr", "lazy", pieces=[(0, 2), (2, 2), (2, 5), (5, 6)],
                                           single_between=True)
    assert_on_trajectory(got, *mf_trajectory(w0, visited, "adam", 0.05), "lazy epochs in pieces around a dense step")
    step, m, v = eng.optimizer_state()
    assert step == len(visited) and not eng._lazy["dirty"]
    # (that the flush leaves the dense sweeps' bits is test_lazy_rows_equal_the_dense_sweeps_bit_for_bit's business)
    assert all(torch.isfinite(t).all() for t in list(m.values()) + list(v.values()))


@pytest.mark.parametrize("engine", ["sharded", "single"])
def test_lazy_adam_at_the_configs3_shard_size(nccl_group, hip_device, engine):
    """One rank's share of BASELINE configs[3] (1.25 M x 125 k rows, dim 128, 65 536 triples per step, Zipf positives)
    with the reference's default optimizer: `dense_opt: "auto"` picks the lazy form at this size, on the row-sharded
    planned path (world 1, C driver) and on MFEngine.  Two epochs of two steps against the numpy oracle's DENSE Adam on
    the compacted problem (the rows the batches touch, renumbered: a never-touched row has zero moments and does not
    move under dense Adam either): epoch loss sums to 1e-5, every touched element inside the legal-trajectory envelope,
    every other row bit-identical with stamp -1, no gradient left behind."""
    import beta_recsys_amd as hp
    from beta_recsys_amd.sharded import ShardedMFEngine

    U, I, D, B, steps, lr = 1_250_000, 125_000, 128, 65536, 2, 0.05
    rng = np.random.default_rng(21)
    pz = 1.0 / np.arange(1, I + 1)
    users = rng.integers(0, U, steps * B)
    pos = rng.permutation(I)[rng.choice(I, steps * B, p=pz / pz.sum())]
    neg = rng.integers(0, I, steps * B)
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer="adam", lr=lr, batch_size=B,
                         loss="bpr", shard_init="local", prefetch_epoch=False), "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    torch.manual_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ShardedMFEngine(cfg) if engine == "sharded" else hp.MFEngine(cfg)
    if engine == "single":
        eng._setup()
    assert eng._lazy is not None, "64 MB of parameters and more take the lazy optimizer"
    m = eng.model
    w0 = m.flat.clone()
    loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a).cuda() for a in (users, pos, neg)), B, shuffle=False)
    sums = []
    with contextlib.redirect_stdout(io.StringIO()):
        for e in range(2):
            out = eng.train_an_epoch(loader, e)
            sums.append(out[0] if engine == "sharded" else eng.writer.scalars[-2][1])
    uu, u_inv = np.unique(users, return_inverse=True)
    ui, i_inv = np.unique(np.concatenate([pos, neg]), return_inverse=True)
    tu, ti = torch.from_numpy(uu).cuda(), torch.from_numpy(ui).cuda()
    ue0, ie0, ub0, ib0, gb0 = m._views(w0)
    wc0 = {"user_emb.weight": ue0[tu].cpu().numpy(), "item_emb.weight": ie0[ti].cpu().numpy(),
           "user_bias.weight": ub0[tu].cpu().numpy(), "item_bias.weight": ib0[ti].cpu().numpy(),
           "global_bias": gb0.cpu().numpy().copy()}
    batches = [(u_inv[k * B:(k + 1) * B], i_inv[:steps * B][k * B:(k + 1) * B], i_inv[steps * B:][k * B:(k + 1) * B])
               for _ in range(2) for k in range(steps)]
    w_ref, env, upd, (sums_ref, _) = mf_trajectory(wc0, batches, "adam", lr, trials=2, with_sums=True)
    w = onp.copy_params(wc0)
    st = onp.new_opt_state(w, "adam")
    for e in range(2):
        tot = sum(onp.mf_train_step(w, st, b, "bpr", "adam", lr)[0] for b in batches[e * steps:(e + 1) * steps])
        assert_scalar_close(sums[e], tot, REL, f"epoch {e} loss sum vs the oracle on the compacted problem")
    ue, ie, ub, ib, gb = m._views(m.flat)
    got = {"user_emb.weight": ue[tu].cpu().numpy(), "item_emb.weight": ie[ti].cpu().numpy(),
           "user_bias.weight": ub[tu].cpu().numpy(), "item_bias.weight": ib[ti].cpu().numpy(),
           "global_bias": gb.cpu().numpy()}
    assert_on_trajectory(got, w_ref, env, upd, f"lazy adam at the configs[3] shard size ({engine})", pool=True)
    for emb, emb0, bias, bias0, ids, n_rows, stamp in ((ue, ue0, ub, ub0, tu, U, eng._lazy["stamp_u"]),
                                                      (ie, ie0, ib, ib0, ti, I, eng._lazy["stamp_i"])):
        idle = torch.ones(n_rows, dtype=torch.bool, device="cuda")
        idle[ids] = False
        assert int(idle.sum()) > 0
        assert torch.equal(emb[idle], emb0[idle]) and torch.equal(bias[idle], bias0[idle]), "an idle row moved"
        assert bool((stamp[idle] == -1).all()) and bool((stamp[~idle] == 2 * steps).all())
    assert float(eng._g_flat.abs().max()) == 0.0 and not eng._lazy["dirty"]


@pytest.mark.parametrize("D,B,grouped", [(64, 1000, True), (128, 4096, True), (100, 777, False), (256, 512, True)])
def test_owned_gradient_kernel_on_local_tables(hip_device, D, B, grouped):
    """hiprec_mf_bpr_grad_owned (the gradient launch of hiprec_mf_epoch_lazy_owned): the dense gradient of one batch
    -- Zipf positives, so rows with one writer (plain stores) and shared rows (atomic adds) both occur; batches grouped
    by positive item as the batcher leaves them, and in drawing order -- against the oracle's, judged like every
    gradient (within 1e-5 of its scale of the exact fp64 gradient + twice the fp32 reference's own distance);
    rows the batch does not name stay exactly zero; nothing is written to the parameters; the clock moves by one."""
    from beta_recsys_amd import _lib
    from beta_recsys_amd.mf import batch_row_ownership, read_stats
    from helpers import assert_grads_as_accurate, float64_oracle, to64

    lib, dev = _lib.load(), torch.device("cuda:0")
    U, I = 5000, 300
    w0 = onp.init_params(U, I, D, seed=7)
    rng = np.random.default_rng(3)
    p = 1.0 / np.arange(1, I + 1)
    users, pos, neg = rng.integers(0, U, B), rng.choice(I, B, p=p / p.sum()), rng.integers(0, I, B)
    if grouped:
        order = np.argsort(pos, kind="stable")
        users, pos, neg = users[order], pos[order], neg[order]
    names = ["user_emb.weight", "item_emb.weight", "user_bias.weight", "item_bias.weight", "global_bias"]
    flat = torch.from_numpy(np.concatenate([np.asarray(w0[k], dtype=np.float32).ravel() for k in names])).to(dev)
    before = flat.clone()
    g = torch.zeros_like(flat)
    tu, tp, tn = (torch.from_numpy(a).to(dev) for a in (users, pos, neg))
    own, total, stride = batch_row_ownership(tu, tp, tn, B, U, I)
    stats = new_stats(lib, _lib, dev)
    scratch = torch.zeros(lib.hiprec_scratch_bytes(0), dtype=torch.uint8, device=dev)
    _lib.check(lib.hiprec_mf_bpr_grad_owned(
        _lib.ptr(flat), _lib.ptr(g), U, I, D, _lib.ptr(tu), _lib.ptr(tp), _lib.ptr(tn), _lib.ptr(own[0]),
        _lib.ptr(own[1]), _lib.ptr(own[2]), _lib.ptr(total), B, 1.0 / B, 0.0, _lib.ptr(stats), _lib.ptr(scratch),
        _lib.stream_ptr(dev)))
    torch.cuda.synchronize()
    assert torch.equal(flat, before) and read_stats(stats).step == 1   # the launch counts the step
    gh = g.cpu().numpy()
    sizes = [U * D, I * D, U, I, 1]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    got = {k: gh[offs[j]:offs[j + 1]].reshape(np.shape(w0[k])) for j, k in enumerate(names[:4])}
    _, _, ref = onp.mf_bpr_grads(w0, users, pos, neg)
    with float64_oracle():
        _, _, exact = onp.mf_bpr_grads(to64(w0), users, pos, neg)
    keys = names[:4]
    assert_grads_as_accurate(got, {k: ref[k] for k in keys}, {k: exact[k] for k in keys},
                             f"owned gradient D={D} B={B}")
    assert gh[offs[4]] == 0.0   # the scalar's gradient travels in the scratch partials
    idle_u = np.setdiff1d(np.arange(U), users)
    idle_i = np.setdiff1d(np.arange(I), np.concatenate([pos, neg]))
    assert not got["user_emb.weight"][idle_u].any() and not got["item_emb.weight"][idle_i].any()
    assert not got["user_bias.weight"].ravel()[idle_u].any() and not got["item_bias.weight"].ravel()[idle_i].any()
