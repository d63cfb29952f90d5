"""Pin oracle/pgmf_numpy.py against golden vectors captured from the real reference's
PairwiseGMFEngine by ``oracle/gen_golden.py --pgmf``.  CPU only."""
import numpy as np
import pytest

from helpers import REL, assert_scalar_close, assert_step_close, assert_tensor_close, copy_state, load_golden
from oracle import pgmf_numpy as onp

CASES = ["pgmf_adam", "pgmf_sgd_clip", "pgmf_rmsprop_init"]
KEYS = onp.KEYS


def pgmf_params(g, prefix):
    return {k: g[f"{prefix}/{k}"].astype(np.float32).copy() for k in KEYS}


def pgmf_opt_state(g, step, opt):
    st = onp.new_opt_state(pgmf_params(g, "w0"), opt)
    st["step"] = step
    if step > 0 and opt == "adam":
        st["exp_avg"], st["exp_avg_sq"] = pgmf_params(g, f"m{step}"), pgmf_params(g, f"v{step}")
    elif step > 0 and opt == "rmsprop":
        st["square_avg"] = pgmf_params(g, f"v{step}")
    return st


def pgmf_band(w_prev, st_prev, g_ref, opt, lr, rel=REL):
    """Forward-error band of one optimizer step for a gradient within rel * its scale of g_ref
    (Adam / RMSprop are ill-conditioned where |g| is not >> eps; see helpers.optimizer_band)."""
    outs = []
    for sign in (+1.0, -1.0):
        w = {k: v.copy() for k, v in w_prev.items()}
        st = copy_state(st_prev)
        g = {k: (g_ref[k] + np.float32(sign * rel * float(np.abs(g_ref[k]).max()))).astype(np.float32)
             for k in KEYS}
        onp.opt_step(w, g, st, opt, lr)
        outs.append(w)
    return {k: np.abs(outs[0][k].astype(np.float64) - outs[1][k].astype(np.float64)) for k in KEYS}


@pytest.mark.parametrize("case", CASES)
def test_numpy_oracle_matches_reference(case):
    """Every step in isolation from the reference's own weights and optimizer state: loss, the CLIPPED
    gradients the reference hands to optimizer.step, the new weights and the new moments."""
    g = load_golden(case)
    n_steps = int(g["meta"][4])
    opt, lr = str(g["optimizer"]), float(g["lr"])
    l2, clip = float(g["l2_lambda"]), float(g["grad_clip"])
    clipped_any = False
    for s in range(n_steps):
        w = pgmf_params(g, f"w{s}")
        st = pgmf_opt_state(g, s, opt)
        loss, grads = onp.pgmf_grads(w, g["users"][s], g["pos"][s], g["neg"][s], l2)
        assert_scalar_close(loss, g["losses"][s], what=f"loss step {s}")
        total = onp.clip_grad_norm(grads, clip)
        clipped_any |= total > clip
        g_ref = pgmf_params(g, f"g{s + 1}")
        for k in KEYS:
            assert_tensor_close(grads[k], g_ref[k], what=f"clipped grad {k} step {s}")
        band = pgmf_band(w, st, g_ref, opt, lr)
        w_prev = {k: v.copy() for k, v in w.items()}
        onp.opt_step(w, grads, st, opt, lr)
        for k in KEYS:
            assert_step_close(w_prev[k], w[k], g[f"w{s + 1}/{k}"], band[k], what=f"weights {k} step {s}")
        nxt = pgmf_opt_state(g, s + 1, opt)
        for name in ("exp_avg", "exp_avg_sq", "square_avg"):
            if name in st:
                for k in KEYS:
                    assert_tensor_close(st[name][k], nxt[name][k], 4e-5, f"{name} {k} step {s}")
    assert clipped_any == (case == "pgmf_sgd_clip"), "the fixture is meant to (not) exercise the clip"


def test_fixtures_exercise_both_relu_branches():
    """A fixture whose scores are all on one side of the relu would leave half the backward unpinned."""
    g = load_golden("pgmf_adam")
    w = pgmf_params(g, "w0")
    _, pre_p = onp.pgmf_scores(w, g["users"][0], g["pos"][0])
    _, pre_n = onp.pgmf_scores(w, g["users"][0], g["neg"][0])
    for pre in (pre_p, pre_n):
        assert (pre > 0).any() and (pre <= 0).any()


def test_trajectory_from_initial_weights():
    """All steps chained from w0 (the oracle's own state carried along) stay on the reference's path."""
    g = load_golden("pgmf_sgd_clip")
    w = pgmf_params(g, "w0")
    st = onp.new_opt_state(w, "sgd")
    for s in range(int(g["meta"][4])):
        loss = onp.pgmf_train_step(w, st, (g["users"][s], g["pos"][s], g["neg"][s]), "sgd", float(g["lr"]),
                                   float(g["l2_lambda"]), float(g["grad_clip"]))
        assert_scalar_close(loss, g["losses"][s], 5e-5, what=f"loss step {s}")
    for k in KEYS:
        assert_tensor_close(w[k], g[f"w3/{k}"], 5e-5, what=f"final {k}")


def test_truncated_normal_restatement_matches_reference_init():
    """oracle truncated_normal fed by the torch RNG in the reference's draw order reproduces the
    reference's initial memories bit for bit (pairwise_gmf.py:35-42)."""
    import torch

    g = load_golden("pgmf_init")
    for tag in ("a", "b"):
        U, I, D, seed = (int(x) for x in g[f"{tag}/meta"])
        torch.manual_seed(seed)
        draw = lambda shape: torch.empty(shape).normal_().numpy()  # noqa: E731
        draw((U, D))                                   # nn.Embedding's own N(0,1) init, overwritten
        um = onp.truncated_normal((U, D), draw, std=0.01)
        draw((I, D))
        im = onp.truncated_normal((I, D), draw, std=0.01)
        assert np.array_equal(um, g[f"{tag}/w/user_memory.weight"])
        assert np.array_equal(im, g[f"{tag}/w/item_memory.weight"])
        assert np.abs(um).max() < 0.02 + 1e-9
