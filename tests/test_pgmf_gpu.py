"""GPU parity tests of PairwiseGMF (the CMN pre-training model): forward + loss +
backward, the gradient-norm clip and the full step vs golden vectors from the real reference's
PairwiseGMFEngine, and vs the numpy oracle at the ML-1M shape."""
import contextlib
import io

import numpy as np
import pytest
import torch

from helpers import REL, assert_sgd_exact  # noqa: F401  (north_star: 1e-5 relative)
from helpers import assert_scalar_close, assert_step_close, assert_tensor_close, load_golden
from oracle import pgmf_numpy as onp
from test_oracle_golden_pgmf import CASES, KEYS, pgmf_band, pgmf_opt_state, pgmf_params

pytestmark = pytest.mark.gpu


def make_engine(U, I, D, B, optimizer="adam", lr=1e-3, l2=1e-4, clip=5.0):
    import beta_recsys_amd as hp

    cfg = {"n_users": U, "n_items": I, "emb_dim": D, "regs": [1e-5], "batch_size": B, "lr": lr,
           "pretrain_l2_lambda": l2, "grad_clip": clip, "neg_count": 4,
           "model": {"device_str": "cuda:0", "optimizer": optimizer, "lr": lr},
           "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    with contextlib.redirect_stdout(io.StringIO()):
        return hp.PairwiseGMFEngine(cfg)


def load_weights(eng, w):
    eng.model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})


def get_weights(eng):
    return {k: v.detach().cpu().numpy() for k, v in eng.model.state_dict().items()}


def load_opt_state(eng, st):
    eng.load_optimizer_state(st["step"], st.get("exp_avg"), st.get("exp_avg_sq", st.get("square_avg")))


def np_grads(grads):
    return {k: v.cpu().numpy() for k, v in grads.items()}


@pytest.mark.parametrize("case", CASES)
def test_step_matches_reference(hip_device, case):
    """Each step from the reference's own weights and optimizer state."""
    g = load_golden(case)
    U, I, D, B, n_steps, seed = (int(x) for x in g["meta"])
    opt, lr, l2, clip = str(g["optimizer"]), float(g["lr"]), float(g["l2_lambda"]), float(g["grad_clip"])
    torch.manual_seed(seed)
    eng = make_engine(U, I, D, B, opt, lr, l2, clip)
    for s in range(n_steps):
        batch = (g["users"][s], g["pos"][s], g["neg"][s])
        w0 = pgmf_params(g, f"w{s}")
        st0 = pgmf_opt_state(g, s, opt)
        load_weights(eng, w0)
        load_opt_state(eng, st0)
        # unclipped gradient vs the oracle (the reference only exposes the clipped one)
        loss_o, g_o = onp.pgmf_grads(w0, *batch, l2)
        loss, grads, norm = eng.backward_only(batch, clip=False)
        assert norm is None
        assert_scalar_close(loss, g["losses"][s], what=f"loss step {s}")
        for k in KEYS:
            assert_tensor_close(np_grads(grads)[k], g_o[k], what=f"raw grad {k} step {s}")
        # clipped gradient vs the reference
        load_opt_state(eng, st0)
        loss, grads, norm = eng.backward_only(batch, clip=True)
        assert_scalar_close(norm, onp.clip_grad_norm(g_o, clip), what=f"total norm step {s}")
        g_ref = pgmf_params(g, f"g{s + 1}")
        for k in KEYS:
            assert_tensor_close(np_grads(grads)[k], g_ref[k], what=f"clipped grad {k} step {s}")
        # the full step
        load_opt_state(eng, st0)
        loss = eng.train_single_batch(batch)
        assert_scalar_close(loss, g["losses"][s], what=f"loss (step) {s}")
        band = pgmf_band(w0, st0, g_ref, opt, lr)
        w1 = get_weights(eng)
        for k in KEYS:
            assert_step_close(w0[k], w1[k], g[f"w{s + 1}/{k}"], band[k], what=f"weights {k} step {s}")
        nxt = pgmf_opt_state(g, s + 1, opt)
        views = {"exp_avg": eng.optimizer.exp_avg, "exp_avg_sq": eng.optimizer.exp_avg_sq}
        for name, ref_name in (("exp_avg", "exp_avg"), ("exp_avg_sq", "exp_avg_sq" if opt == "adam" else "square_avg")):
            if views[name] is None:
                continue
            got = {k: v.cpu().numpy() for k, v in eng.model.views(views[name]).items()}
            for k in KEYS:
                assert_tensor_close(got[k], nxt[ref_name][k], REL if name == "exp_avg" else 2 * REL,
                                    f"{name} {k} step {s}")   # linear / quadratic in the gradient
        assert float(eng._g_flat.abs().max()) == 0.0, "the optimizer sweep leaves the gradient cleared"


def test_trajectory_matches_reference(hip_device):
    """Three chained SGD steps with the clip active on every one."""
    g = load_golden("pgmf_sgd_clip")
    U, I, D, B, n_steps, _ = (int(x) for x in g["meta"])
    eng = make_engine(U, I, D, B, "sgd", float(g["lr"]), float(g["l2_lambda"]), float(g["grad_clip"]))
    load_weights(eng, pgmf_params(g, "w0"))
    for s in range(n_steps):
        loss = eng.train_single_batch((g["users"][s], g["pos"][s], g["neg"][s]))
        assert_scalar_close(loss, g["losses"][s], REL, what=f"loss step {s}")
    w = get_weights(eng)
    # plain SGD has no conditioning problem: every element within 1e-5 of the trajectory's total update (+ 4 ulp)
    assert_sgd_exact(w, pgmf_params(g, f"w{n_steps}"), pgmf_params(g, "w0"), "final weights")


def test_forward_scores(hip_device):
    g = load_golden("pgmf_adam")
    U, I, D, B = (int(x) for x in g["meta"][:4])
    eng = make_engine(U, I, D, B)
    w = pgmf_params(g, "w0")
    load_weights(eng, w)
    pos, neg = eng.model(g["users"][0], g["pos"][0], g["neg"][0])
    assert tuple(pos.shape) == (B, 1) and tuple(neg.shape) == (B, 1)
    assert_tensor_close(pos.cpu().numpy()[:, 0], onp.pgmf_scores(w, g["users"][0], g["pos"][0])[0], what="pos")
    assert_tensor_close(neg.cpu().numpy()[:, 0], onp.pgmf_scores(w, g["users"][0], g["neg"][0])[0], what="neg")
    with pytest.raises(IndexError):
        eng.model([0, U], [0, 0], [0, 0])


@pytest.mark.parametrize("dim,optimizer", [(64, "adam"), (128, "sgd"), (256, "rmsprop"), (20, "adam")])
def test_ml1m_shape_vs_oracle(hip_device, dim, optimizer):
    """BASELINE C2 table shape (6040 x 3706), batch 4096 with Zipf items: loss, gradients, norm and two
    full steps vs the oracle; clip chosen so that it is active."""
    U, I, B = 6040, 3706, 4096
    rng = np.random.default_rng(dim)
    torch.manual_seed(dim)
    lr, l2 = (0.05 if optimizer == "sgd" else 1e-3), 1e-3
    eng = make_engine(U, I, dim, B, optimizer, lr, l2, clip=1e-3)
    w = get_weights(eng)
    w["user_memory.weight"] *= 40.0
    w["item_memory.weight"] *= 40.0
    load_weights(eng, w)
    perm = rng.permutation(I)
    zipf = lambda n: perm[np.minimum((rng.zipf(1.2, n) - 1), I - 1)]  # noqa: E731
    st = onp.new_opt_state(w, optimizer)
    for step in range(2):
        batch = (rng.integers(0, U, B), zipf(B), rng.integers(0, I, B))
        loss_o, g_o = onp.pgmf_grads(w, *batch, l2)
        if step == 0:
            loss, grads, norm = eng.backward_only(batch, clip=True)
            eng.load_optimizer_state(0)
            g_c = {k: v.copy() for k, v in g_o.items()}
            total = onp.clip_grad_norm(g_c, 1e-3)
            assert total > 1e-3, "the clip is meant to be active"
            assert_scalar_close(loss, loss_o, what="loss")
            assert_scalar_close(norm, total, what="total norm")
            got = np_grads(grads)
            for k in KEYS:
                assert_tensor_close(got[k], g_c[k], what=f"clipped grad {k}")
            # size-independent property: a clipped gradient has norm max_norm * total / (total + 1e-6)
            flat = np.concatenate([got[k].ravel() for k in KEYS]).astype(np.float64)
            assert_scalar_close(np.sqrt((flat * flat).sum()), 1e-3 * total / (total + 1e-6), what="clipped norm")
            # rows no triple touched keep a zero gradient
            untouched = np.setdiff1d(np.arange(U), batch[0])
            assert not got["user_memory.weight"][untouched].any()
        w_prev = {k: v.copy() for k, v in w.items()}
        st_prev = {k: ({kk: vv.copy() for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in st.items()}
        loss_o = onp.pgmf_train_step(w, st, batch, optimizer, lr, l2, 1e-3)
        loss = eng.train_single_batch(batch)
        assert_scalar_close(loss, loss_o, what=f"loss step {step}")
        g_c = onp.pgmf_grads(w_prev, *batch, l2)[1]
        onp.clip_grad_norm(g_c, 1e-3)
        band = pgmf_band(w_prev, st_prev, g_c, optimizer, lr)
        got = get_weights(eng)
        for k in KEYS:
            assert_step_close(w_prev[k], got[k], w[k], band[k], what=f"weights {k} step {step}")
        if optimizer == "sgd" and step == 0:
            untouched = np.setdiff1d(np.arange(U), batch[0])
            assert np.array_equal(got["user_memory.weight"][untouched], w_prev["user_memory.weight"][untouched])
        load_weights(eng, w)  # keep the two trajectories on the same point


def test_clip_grad_norm_entry_point(hip_device):
    """hiprec_clip_grad_norm stand-alone: inactive clip leaves g bit-identical, active clip scales it;
    odd lengths exercise the scalar tail; n = 0 is accepted."""
    from beta_recsys_amd import _lib

    lib = _lib.load()
    st = _lib.stream_ptr(hip_device)
    ws = torch.zeros(lib.hiprec_clip_workspace_bytes() // 8, dtype=torch.float64, device="cuda")
    rng = np.random.default_rng(5)
    for n in (1, 3, 4, 1027, 2_000_003):
        x = rng.standard_normal(n).astype(np.float32)
        total = np.sqrt((x.astype(np.float64) ** 2).sum())
        g = torch.from_numpy(x).cuda()
        _lib.check(lib.hiprec_clip_grad_norm(_lib.ptr(g), n, float(2 * total), _lib.ptr(ws), ws.numel() * 8, st))
        assert np.array_equal(g.cpu().numpy(), x)
        assert_scalar_close(float(ws[0]), total, 2e-6, f"norm n={n}")
        assert float(ws[1]) == 1.0
        _lib.check(lib.hiprec_clip_grad_norm(_lib.ptr(g), n, float(total / 4), _lib.ptr(ws), ws.numel() * 8, st))
        coef = np.float32(np.float32(total / 4) / (np.float32(total) + np.float32(1e-6)))
        assert_tensor_close(g.cpu().numpy(), x * coef, 1e-6, f"scaled n={n}")
    _lib.check(lib.hiprec_clip_grad_norm(None, 0, 1.0, _lib.ptr(ws), ws.numel() * 8, st))
    assert float(ws[0]) == 0.0


@pytest.mark.parametrize("kind,name", [(0, "sgd"), (1, "adam"), (2, "rmsprop")])
def test_clip_riding_in_the_optimizer_sweep_leaves_the_same_bits(hip_device, kind, name):
    """hiprec_clip_opt_dense_step (two launches: sums of squares, then a sweep that scales and steps) against
    hiprec_clip_grad_norm + hiprec_opt_dense_step (three): w, m, v, the cleared g and the workspace's (total_norm, coef)
    BIT FOR BIT -- with an active clip, an inactive one, a NaN in the gradient (torch poisons every element) and lengths
    that exercise the scalar tail; two steps, so the second starts from non-zero moments."""
    from beta_recsys_amd import _lib
    from beta_recsys_amd.mf import _new_stats

    lib = _lib.load()
    st = _lib.stream_ptr(hip_device)
    rng = np.random.default_rng(11 + kind)
    for n, factor, poison in ((2_000_003, 0.25, False), (1027, 4.0, False), (4096, 0.5, True), (5, 0.3, False)):
        x = [rng.standard_normal(n).astype(np.float32) for _ in range(3)]
        if poison:
            x[2][7] = np.nan
        total = float(np.sqrt((x[1].astype(np.float64) ** 2).sum()))
        out = []
        for fused in (False, True):
            w = torch.from_numpy(x[0]).cuda()
            m, v = torch.zeros_like(w), torch.zeros_like(w)
            ws = torch.zeros(lib.hiprec_clip_workspace_bytes() // 8, dtype=torch.float64, device="cuda")
            stats = _new_stats(w.device)
            _lib.check(lib.hiprec_stats_reset(_lib.ptr(stats), 0.9, 0.999, st))
            norms = []
            for step in (1, 2):
                g = torch.from_numpy(x[step]).cuda()
                _lib.check(lib.hiprec_stats_advance_step(_lib.ptr(stats), st))
                args = (kind, _lib.ptr(w), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), n, 0.01, 0.9, 0.999, 1e-8, _lib.ptr(stats),
                        None, -1)
                if fused:
                    _lib.check(lib.hiprec_clip_opt_dense_step(*args, float(factor * total), _lib.ptr(ws), ws.numel() * 8, st))
                else:
                    _lib.check(lib.hiprec_clip_grad_norm(_lib.ptr(g), n, float(factor * total), _lib.ptr(ws),
                                                         ws.numel() * 8, st))
                    _lib.check(lib.hiprec_opt_dense_step(*args, st))
                norms.append(ws[:2].cpu().numpy().copy())
                assert not g.cpu().numpy().any() or poison   # the sweep leaves the gradient cleared
            out.append([t.cpu().numpy() for t in (w, m, v)] + norms)
        for a, b, what in zip(out[0], out[1], ("w", "m", "v", "norm step 1", "norm step 2")):
            assert np.array_equal(a, b, equal_nan=True), f"{name} n={n} factor={factor}: {what} differs"
        assert (out[1][3][1] < 1.0) == (factor < 1.0)
        if poison:
            assert np.isnan(out[1][0]).all()   # clip_grad_norm_ with a NaN norm scales everything by NaN


def test_bad_indices_and_batches(hip_device):
    eng = make_engine(12, 9, 8, 4)
    with pytest.raises(IndexError):
        eng.train_single_batch(([0, 12], [1, 2], [3, 4]))
    with pytest.raises(IndexError):
        eng.train_single_batch(([0, 1], [1, 9], [3, 4]))
    with pytest.raises(IndexError):
        eng.train_single_batch(([0, 1], [1, 2], [3, -1]))
    with pytest.raises(ValueError):
        eng.train_single_batch(([], [], []))
    with pytest.raises(ValueError):
        eng.train_single_batch(([0, 1], [1], [3, 4]))
    # the engine keeps working after an error
    assert np.isfinite(eng.train_single_batch(([0, 1], [1, 2], [3, 4])))


def test_train_an_epoch_uses_cmn_loader(hip_device):
    """pairwise_gmf.py:118-142: batches from train_loader.cmn_train_loader(batch_size, False, neg_count)
    as [B,3] uint32 arrays; the epoch sum goes to the writer, the LAST batch's loss is printed."""
    U, I, D, B = 50, 40, 16, 32
    rng = np.random.default_rng(3)
    batches = [np.stack([rng.integers(0, U, n), rng.integers(0, I, n), rng.integers(0, I, n)], 1).astype(np.uint32)
               for n in (B, B, B, 7)]

    class Loader:
        n_train = 100
        calls = []

        def cmn_train_loader(self, batch_size, neighborhood, neg_count):
            self.calls.append((batch_size, neighborhood, neg_count))
            return iter(batches)

    eng = make_engine(U, I, D, B, "adam", 1e-2)
    w = get_weights(eng)
    for k in ("user_memory.weight", "item_memory.weight"):
        w[k] *= 50.0
    load_weights(eng, w)
    st = onp.new_opt_state(w, "adam")
    w_start = {k: v.copy() for k, v in w.items()}
    ref = [onp.pgmf_train_step(w, st, (b[:, 0].astype(np.int64), b[:, 1].astype(np.int64), b[:, 2].astype(np.int64)),
                               "adam", 1e-2, 1e-4, 5.0) for b in batches]
    out = io.StringIO()
    loader = Loader()
    with contextlib.redirect_stdout(out):
        eng.train_an_epoch(loader, 3)
    assert loader.calls == [(B, False, 4)]
    tag, total, epoch = eng.writer.scalars[-1]
    assert (tag, epoch) == ("model/loss", 3)
    assert_scalar_close(total, sum(ref), REL, "epoch loss sum")
    printed = float(out.getvalue().strip().rsplit("Loss ", 1)[1])
    assert_scalar_close(printed, ref[-1], REL, "printed last loss")
    # every element inside the legal-trajectory envelope (oracle runs with each gradient moved by 1e-5 of its scale
    # BEFORE the clip, as an implementation's own gradient would be)
    from helpers import assert_on_trajectory, oracle_trajectory

    def clipped_grads(wc, b):
        _, gr = onp.pgmf_grads(wc, b[:, 0].astype(np.int64), b[:, 1].astype(np.int64), b[:, 2].astype(np.int64), 1e-4)
        return gr

    def clip_then_step(wc, gr, state):
        onp.clip_grad_norm(gr, 5.0)
        onp.opt_step(wc, gr, state, "adam", 1e-2)

    w_ref, env, upd = oracle_trajectory(w_start, batches, clipped_grads, clip_then_step,
                                        lambda wc: onp.new_opt_state(wc, "adam"))
    assert all(np.array_equal(w_ref[k], w[k]) for k in KEYS)
    assert_on_trajectory(get_weights(eng), w_ref, env, upd, "PairwiseGMF epoch")
