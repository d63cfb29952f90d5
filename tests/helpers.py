"""Shared helpers for the parity tests."""
import contextlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ("global_bias", "user_emb.weight", "item_emb.weight", "user_bias.weight", "item_bias.weight")

# north_star tolerance: 1e-5 relative on fp32 loss / grad (BASELINE.json); integer gathers bit-exact
REL = 1e-5
EPS32 = float(np.finfo(np.float32).eps)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def params(gold, prefix):
    return {k: gold[f"{prefix}/{k}"].astype(np.float32).copy() for k in KEYS}


def assert_scalar_close(got, ref, rel=REL, what=""):
    assert abs(float(got) - float(ref)) <= rel * abs(float(ref)) + 1e-12, (
        f"{what}: got {got!r}, reference {ref!r}, rel err {abs(got - ref) / max(abs(ref), 1e-30):.3e}")


def grad_scale_floor(key, batch):
    """Natural magnitude of the TERMS a bias gradient sums.  d(loss)/d(bias) adds per-sample terms
    of size <= 0.25/B with opposite signs (positive vs negative item), so the result can be orders
    of magnitude below its terms and its fp32 rounding error is relative to the terms, not to the
    cancelled result: global_bias sums 2B of them (two partial sums of ~0.125 each)."""
    if key == "global_bias":
        return 0.125
    if key.endswith("bias.weight"):
        return 0.25 / batch
    return 0.0


def assert_tensor_close(got, ref, rel=REL, what="", scale_floor=0.0):
    """max|got-ref| <= rel * max(max|ref|, scale_floor)  (error relative to the tensor's scale)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    scale = max(np.abs(ref).max() if ref.size else 0.0, scale_floor)
    err = np.abs(got - ref).max() if ref.size else 0.0
    assert err <= rel * scale + 1e-30, f"{what}: max err {err:.3e} > {rel:g} * scale {scale:.3e}"


def assert_as_accurate_as_reference(got, ref, exact, rel=REL, what="", ref_factor=2.0, scale_floor=0.0):
    """For quantities whose fp32 evaluation is ill-conditioned (NGCF: the last hop's row normalisation
    cancels): ``exact`` is the oracle evaluated in fp64, ``ref`` the reference's own fp32 result.
    ``got`` must be within rel * scale of the exact value, plus ``ref_factor`` (twice) the distance the reference
    itself is from it — i.e. as accurate as the reference, not bit-compatible with its rounding.  (Round 2 passed 3
    for NGCF's bias gradients, column sums then accumulated with run-order atomics; they are two-level fixed-order
    sums now and meet 2 like everything else.)"""
    got, ref, exact = (np.asarray(a, dtype=np.float64) for a in (got, ref, exact))
    assert got.shape == exact.shape, f"{what}: shape {got.shape} vs {exact.shape}"
    scale = max(np.abs(exact).max() if exact.size else 0.0, scale_floor)
    ref_err = np.abs(ref - exact).max() if exact.size else 0.0
    got_err = np.abs(got - exact).max() if exact.size else 0.0
    assert got_err <= rel * scale + ref_factor * ref_err + 1e-30, (
        f"{what}: err vs exact {got_err:.3e} > {rel:g} * scale {scale:.3e} + {ref_factor:g} * reference's own error "
        f"{ref_err:.3e}")


def assert_update_close(w0, got, ref, rel=REL, what=""):
    """Compare the UPDATE (w_new - w_old): tolerance rel * max|update| plus a few ulps of the
    weights themselves (the update is added to an fp32 weight)."""
    w0, got, ref = (np.asarray(a, dtype=np.float64) for a in (w0, got, ref))
    upd_scale = np.abs(ref - w0).max()
    w_scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    tol = rel * upd_scale + 4 * EPS32 * w_scale
    assert err <= tol, (f"{what}: max err {err:.3e} > tol {tol:.3e} "
                        f"(update scale {upd_scale:.3e}, weight scale {w_scale:.3e})")


def golden_opt_state(gold, step, opt):
    """Optimizer state of the reference AFTER `step` steps, in oracle/mf_numpy.py's format."""
    from oracle import mf_numpy as onp

    w_like = params(gold, "w0")
    st = onp.new_opt_state(w_like, opt)
    st["step"] = step
    if step > 0:
        if opt == "adam":
            st["exp_avg"] = params(gold, f"m{step}")
            st["exp_avg_sq"] = params(gold, f"v{step}")
        elif opt == "rmsprop":
            st["square_avg"] = params(gold, f"v{step}")
    return st


def copy_state(st):
    return {k: ({kk: vv.copy() for kk, vv in v.items()} if isinstance(v, dict) else v)
            for k, v in st.items()}


def optimizer_band(w_prev, st_prev, g_ref, opt, lr, batch, rel=REL):
    """Elementwise forward-error band of one optimizer step for a gradient that is within the
    stated tolerance (rel * its scale) of the reference gradient.

    Adam / RMSprop divide by sqrt(v)+eps: where |g| is not >> eps = 1e-8 the update is
    ill-conditioned (d update / d g ~ lr/eps), so a gradient that is correct to 1e-5 relative can
    legitimately move such an element by far more than 1e-5 of the update scale.  The band is
    |step(g + d) - step(g - d)| with d = rel * scale(g), evaluated with the oracle."""
    from oracle import mf_numpy as onp

    outs = []
    for sign in (+1.0, -1.0):
        w = onp.copy_params(w_prev)
        st = copy_state(st_prev)
        g = {}
        for k in KEYS:
            scale = max(float(np.abs(g_ref[k]).max()), grad_scale_floor(k, batch))
            g[k] = (g_ref[k] + np.float32(sign * rel * scale)).astype(np.float32)
        onp.opt_step(w, g, st, opt, lr)
        outs.append(w)
    return {k: np.abs(outs[0][k].astype(np.float64) - outs[1][k].astype(np.float64)) for k in KEYS}


def assert_step_close(w_prev, got, ref, band, rel=REL, what=""):
    """|got - ref| <= rel*max|ref - w_prev| + 4 ulp(max|w|) + band   (elementwise)."""
    w_prev, got, ref = (np.asarray(a, dtype=np.float64) for a in (w_prev, got, ref))
    tol = rel * np.abs(ref - w_prev).max() + 4 * EPS32 * np.abs(ref).max() + band
    err = np.abs(got - ref)
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())} elements out of tolerance, worst err "
                           f"{err[bad].max():.3e} vs tol {tol[bad][np.argmax(err[bad])]:.3e}")


def legal_trajectory_envelope(gold, n_steps, opt, lr, batch, loss="bpr", reg_coef=0.0, trials=8, rel=REL, seed=0):
    """Elementwise envelope of where a CORRECT implementation may end up after `n_steps` chained steps.

    north_star allows every gradient to be off by rel = 1e-5 of its tensor's scale.  Adam / RMSprop turn such an error
    into anything up to ~lr for elements whose gradient is ~eps, and the moments carry it into the next steps, so a
    per-step tolerance cannot be summed into a multi-step one.  Instead the oracle itself is run `trials` times with
    every gradient element moved by +-rel * scale (random signs): each run is a legal trajectory.  Returns, per
    parameter tensor, max over the runs of |w_run - w_reference|."""
    from oracle import mf_numpy as onp

    rng = np.random.default_rng(seed)
    env = {k: np.zeros(gold[f"w0/{k}"].shape, dtype=np.float64) for k in KEYS}
    for _ in range(trials):
        w = params(gold, "w0")
        st = onp.new_opt_state(w, opt)
        for s in range(n_steps):
            batch_s = tuple(gold[k][s] for k in ("users", "items_a", "third"))
            fn = onp.mf_bpr_grads if loss == "bpr" else onp.mf_bce_grads
            _, _, g = fn(w, batch_s[0], batch_s[1], batch_s[2], reg_coef)
            for k in KEYS:
                scale = max(float(np.abs(g[k]).max()), grad_scale_floor(k, batch))
                g[k] = (g[k] + (rng.choice([-1.0, 1.0], size=g[k].shape) * rel * scale).astype(np.float32)).astype(np.float32)
            onp.opt_step(w, g, st, opt, lr)
        for k in KEYS:
            env[k] = np.maximum(env[k], np.abs(w[k].astype(np.float64) - gold[f"w{n_steps}/{k}"]))
    return env


def oracle_trajectory(w0, batches, grad_fn, step_fn, new_state, floor_fn=None, trials=8, rel=REL, seed=0):
    """The generalisation of :func:`legal_trajectory_envelope` over an oracle-step callback (any model, any data):
    ``grad_fn(w, batch) -> {name: gradient}``, ``step_fn(w, g, state)`` updates ``w`` / ``state`` in place,
    ``new_state(w)`` makes the optimizer state, ``floor_fn(name, batch)`` the scale floor of a gradient tensor.

    Returns ``(w_ref, env, upd)``: the oracle's own end point, the elementwise envelope max |w_run - w_ref| over
    `trials` runs in which every gradient element is moved by +-rel of its tensor's scale (what north_star allows an
    implementation to differ by), and per tensor the largest single-step update of the reference run."""
    rng = np.random.default_rng(seed)

    def run(perturb):
        w = {k: np.array(v, dtype=np.float32) for k, v in w0.items()}
        st = new_state(w)
        upd = {k: 0.0 for k in w}
        for b in batches:
            g = grad_fn(w, b)
            if perturb:
                for k in g:
                    scale = max(float(np.abs(g[k]).max()) if g[k].size else 0.0, floor_fn(k, b) if floor_fn else 0.0)
                    r = rel[k] if isinstance(rel, dict) else rel   # per tensor where the reference itself is less exact
                    g[k] = (g[k] + (rng.choice([-1.0, 1.0], size=g[k].shape) * r * scale).astype(np.float32)
                            ).astype(np.float32)
            before = {k: v.copy() for k, v in w.items()}
            step_fn(w, g, st)
            for k in w:
                upd[k] = max(upd[k], float(np.abs(w[k].astype(np.float64) - before[k]).max()) if w[k].size else 0.0)
        return w, upd

    w_ref, upd = run(False)
    env = {k: np.zeros(v.shape, dtype=np.float64) for k, v in w_ref.items()}
    for _ in range(trials):
        w, _ = run(True)
        for k in env:
            env[k] = np.maximum(env[k], np.abs(w[k].astype(np.float64) - w_ref[k]))
    return w_ref, env, upd


def assert_on_trajectory(got, w_ref, env, upd, what="", rel=REL, env_factor=2.0, pool=False):  # noqa: D401
    """EVERY element within env_factor x the legal envelope + rel of the largest update + 4 ulp of the weights: no
    allowance for a fraction of outliers.  ``pool``: the envelope of a tensor is the LARGEST deviation any of its
    elements showed in any legal run -- for long trajectories (hundreds of Adam steps), where which element drifts is
    itself chaotic and an elementwise maximum over a handful of runs under-covers the next run."""
    for k in w_ref:
        ref = w_ref[k].astype(np.float64)
        g = np.asarray(got[k], dtype=np.float64).reshape(ref.shape)
        r = rel[k] if isinstance(rel, dict) else rel
        e = np.full(ref.shape, env[k].max() if env[k].size else 0.0) if pool else env[k]
        bound = env_factor * e + r * upd[k] + 4 * EPS32 * (np.abs(ref).max() if ref.size else 0.0)
        bad = np.abs(g - ref) > bound
        assert not bad.any(), (f"{what} {k}: {int(bad.sum())} of {bad.size} elements off the reference trajectory, "
                               f"worst {np.abs(g - ref)[bad].max():.3e} vs bound {bound[bad].min():.3e}")


def assert_sgd_exact(got, w_ref, w0, what="", rel=REL, lr=None, batch=None):
    """Plain SGD has no conditioning problem: every element within rel of the largest update + 4 ulp, zero outliers.
    With lr / batch given, the update scale of a bias tensor is at least lr x the scale of the TERMS its gradient
    sums (grad_scale_floor: they cancel, the rounding error does not)."""
    for k in w_ref:
        ref = np.asarray(w_ref[k], dtype=np.float64)
        g = np.asarray(got[k], dtype=np.float64).reshape(ref.shape)
        upd = np.abs(ref - np.asarray(w0[k], dtype=np.float64).reshape(ref.shape)).max() if ref.size else 0.0
        if lr is not None:
            upd = max(upd, lr * grad_scale_floor(k, batch))
        tol = rel * upd + 4 * EPS32 * (np.abs(ref).max() if ref.size else 0.0)
        err = np.abs(g - ref).max() if ref.size else 0.0
        assert err <= tol, f"{what} {k}: max err {err:.3e} > {tol:.3e} (update scale {upd:.3e})"


def mf_trajectory(w0, batches, opt, lr, reg_coef=0.0, loss="bpr", with_sums=False, **kw):
    """oracle_trajectory for BPR / BCE MF on a list of (users, items, third) batches.  ``with_sums``: also return
    ``(sums_ref, sums_env)``, the reference run's (loss sum, regularizer sum) over the batches and the largest
    deviation of the perturbed runs' sums from them -- the legal spread of an epoch's scalars."""
    from oracle import mf_numpy as onp

    fn = onp.mf_bpr_grads if loss == "bpr" else onp.mf_bce_grads
    scalars = []

    def grad_fn(w, b):
        lo, rg, g = fn(w, b[0], b[1], b[2], reg_coef)
        scalars.append((float(lo), float(rg)))
        return g

    out = oracle_trajectory(
        w0, batches, grad_fn, lambda w, g, st: onp.opt_step(w, g, st, opt, lr), lambda w: onp.new_opt_state(w, opt),
        lambda k, b: grad_scale_floor(k, len(b[0])), **kw)
    if not with_sums:
        return out
    per_run = np.asarray(scalars, dtype=np.float64).reshape(-1, len(batches), 2).sum(1)   # run 0 = the reference run
    spread = np.abs(per_run[1:] - per_run[0]).max(0) if len(per_run) > 1 else np.zeros(2)
    return out + ((per_run[0], spread),)


def assert_mf_end_state(got, w0, batches, opt, lr, what="", loss="bpr", reg_coef=0.0, ref=None, trials=8, pool=False):
    """The weights after training MF on `batches` from `w0`: plain SGD every element within 1e-5 of the update
    (no conditioning problem, zero outliers), Adam / RMSprop every element inside the legal-trajectory envelope.
    `ref`: the reference's own end point when a golden holds it (default: the oracle's).  Returns the trajectory
    ``(w_ref, env, upd)`` for callers that check a second run against the same oracle runs."""
    traj = mf_trajectory(w0, batches, opt, lr, reg_coef, loss, trials=0 if opt == "sgd" else trials)
    check_mf_end_state(got, w0, batches, opt, lr, traj, what, ref, pool)
    return traj


def check_mf_end_state(got, w0, batches, opt, lr, traj, what="", ref=None, pool=False):
    w_ref, env, upd = traj
    ref = w_ref if ref is None else ref
    if opt == "sgd":
        assert_sgd_exact(got, ref, w0, what, lr=lr, batch=min(len(b[0]) for b in batches))
    else:
        assert_on_trajectory(got, ref, env, upd, what, pool=pool)


def ncf_trajectory(w0, batches, kind, opt, lr, **kw):
    """oracle_trajectory for the NCF family on (users, items, ratings) batches."""
    from oracle import ncf_numpy as onc

    return oracle_trajectory(w0, batches, lambda w, b: onc.ncf_grads(w, b[0], b[1], b[2], kind)[1],
                             lambda w, g, st: onc.opt_step(w, g, st, opt, lr), lambda w: onc.new_opt_state(w, opt), **kw)


def assert_ncf_end_state(got, w0, batches, kind, opt, lr, what="", ref=None):
    w_ref, env, upd = ncf_trajectory(w0, batches, kind, opt, lr, trials=0 if opt == "sgd" else 8)
    ref = w_ref if ref is None else ref
    if opt == "sgd":
        assert_sgd_exact(got, ref, w0, what)
    else:
        assert_on_trajectory(got, ref, env, upd, what)


@contextlib.contextmanager
def float64_oracle(*modules):
    """Evaluate oracle modules in fp64 (the EXACT value of what they restate, up to 1e-16): the restatements spell
    their working precision as the module global ``F32``; inside this context it means float64 (pass the weights
    through :func:`to64`).  The yardstick of :func:`assert_as_accurate_as_reference`: an fp32 gradient is a sum of
    B terms in SOME order; the reference's own order leaves it ~1e-5 of its scale away from the exact sum, and an
    implementation that sums in another order cannot be held to more than the same distance."""
    from oracle import mf_numpy

    mods = (mf_numpy,) + tuple(modules)
    saved = [m.F32 for m in mods]
    for m in mods:
        m.F32 = np.float64
    try:
        yield
    finally:
        for m, f in zip(mods, saved):
            m.F32 = f


def to64(w):
    return {k: np.asarray(v, dtype=np.float64) for k, v in w.items()}


def assert_grads_as_accurate(got, ref, exact, what="", floor_fn=None, rel=REL, ref_factor=2.0):
    """Every gradient tensor within rel of its scale of the EXACT (fp64) gradient + twice the distance the fp32
    reference itself is from it; prints nothing, raises with the measured reference error in the message."""
    for k in ref:
        g = got[k].cpu().numpy() if hasattr(got[k], "cpu") else np.asarray(got[k])
        assert_as_accurate_as_reference(g.reshape(np.shape(exact[k])), ref[k], exact[k], rel, f"{what} {k}", ref_factor,
                                        floor_fn(k) if floor_fn else 0.0)
