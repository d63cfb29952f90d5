"""Pin oracle/ngcf_numpy.py against golden vectors captured from the real reference's NGCFEngine by
``oracle/gen_golden.py --ngcf`` (message-dropout masks included).  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import REL, assert_as_accurate_as_reference, assert_scalar_close, assert_step_close
from helpers import assert_tensor_close, copy_state, load_golden
from oracle import ngcf_numpy as onp

CASES = ["ngcf_adam", "ngcf_sgd_widths", "ngcf_rmsprop_d64"]


def ngcf_adj(g):
    N = int(g["meta"][0] + g["meta"][1])
    a = sp.csr_matrix((g["adj_val"], (g["adj_row"], g["adj_col"])), shape=(N, N))
    a.sort_indices()
    return a


def ngcf_keys(g):
    return onp.keys(int(g["meta"][3]))


def ngcf_params(g, prefix):
    return {k: g[f"{prefix}/{k}"].astype(np.float32).copy() for k in ngcf_keys(g)}


def ngcf_batch(g, s):
    return g[f"b{s}/users"], g[f"b{s}/pos"], g[f"b{s}/neg"]


def ngcf_masks(g, s):
    """Per hop the [N, d] keep mask of step s (None where the hop has no dropout)."""
    N = int(g["meta"][0] + g["meta"][1])
    out = []
    for l, p in enumerate(g["mess_dropout"]):
        d = int(g["layers"][l])
        out.append(None if p == 0 else np.unpackbits(g[f"mask{s}/{l}"], axis=1)[:, :d].astype(bool).reshape(N, d))
    return out


def ngcf_opt_state(g, step, opt):
    st = onp.new_opt_state(ngcf_params(g, "w0"), opt)
    st["step"] = step
    if step > 0 and opt == "adam":
        st["exp_avg"], st["exp_avg_sq"] = ngcf_params(g, f"m{step}"), ngcf_params(g, f"v{step}")
    elif step > 0 and opt == "rmsprop":
        st["square_avg"] = ngcf_params(g, f"v{step}")
    return st


def ngcf_grad_rel(g_ref, exact, rel=REL):
    """Per tensor: the relative gradient error assert_as_accurate_as_reference admits."""
    return {k: rel + 2.0 * float(np.abs(g_ref[k] - exact[k]).max()) / float(np.abs(exact[k]).max()) for k in g_ref}


def ngcf_band(w_prev, st_prev, g_ref, opt, lr, rel=REL):
    """Forward-error band of one optimizer step for gradients within rel (a float or a per-tensor dict)
    of their scale."""
    if not isinstance(rel, dict):
        rel = {k: rel for k in w_prev}
    outs = []
    for sign in (+1.0, -1.0):
        w = {k: v.copy() for k, v in w_prev.items()}
        st = copy_state(st_prev)
        g = {k: (g_ref[k] + np.float32(sign * rel[k] * float(np.abs(g_ref[k]).max()))).astype(np.float32) for k in w}
        onp.opt_step(w, g, st, opt, lr)
        outs.append(w)
    return {k: np.abs(outs[0][k].astype(np.float64) - outs[1][k].astype(np.float64)) for k in w_prev}


@pytest.mark.parametrize("case", CASES)
def test_numpy_oracle_matches_reference(case):
    """Every step in isolation from the reference's own weights, optimizer state and dropout masks."""
    g = load_golden(case)
    B, n_steps = int(g["meta"][4]), int(g["meta"][5])
    opt, lr, decay = str(g["optimizer"]), float(g["lr"]), float(g["decay"])
    adj, drop = ngcf_adj(g), [float(p) for p in g["mess_dropout"]]
    for s in range(n_steps):
        w = ngcf_params(g, f"w{s}")
        st = ngcf_opt_state(g, s, opt)
        loss, grads = onp.ngcf_grads(w, adj, *ngcf_batch(g, s), decay, B, ngcf_masks(g, s), drop)
        assert_scalar_close(loss, g["losses"][s], what=f"loss step {s}")
        g_ref = ngcf_params(g, f"g{s + 1}")
        _, exact = onp.ngcf_grads(w, adj, *ngcf_batch(g, s), decay, B, ngcf_masks(g, s), drop, dt=np.float64)
        for k in w:
            assert_as_accurate_as_reference(grads[k], g_ref[k], exact[k], what=f"grad {k} step {s}")
        band = ngcf_band(w, st, g_ref, opt, lr, rel=ngcf_grad_rel(g_ref, exact))
        w_prev = {k: v.copy() for k, v in w.items()}
        onp.opt_step(w, grads, st, opt, lr)
        for k in w:
            assert_step_close(w_prev[k], w[k], g[f"w{s + 1}/{k}"], band[k], what=f"weights {k} step {s}")
        nxt = ngcf_opt_state(g, s + 1, opt)
        grel = ngcf_grad_rel(g_ref, exact)
        for name in ("exp_avg", "exp_avg_sq", "square_avg"):
            if name in st:
                for k in w:  # moments are linear (m) / quadratic (v) in the gradient
                    assert_tensor_close(st[name][k], nxt[name][k], 4e-5 + 2 * grel[k], f"{name} {k} step {s}")


def test_state_dict_order_and_short_batch_regulariser():
    g = load_golden("ngcf_adam")
    stored = [k[len("w0/"):] for k in g if k.startswith("w0/")]
    assert stored == list(ngcf_keys(g)), "npz keeps the reference's state_dict order"
    # ngcf.py:189: the regulariser divides by the CONFIGURED batch size (24), the BPR mean by the real one (9)
    assert list(g["batch_lens"]) == [24, 24, 9]
    w, adj, drop = ngcf_params(g, "w2"), ngcf_adj(g), [float(p) for p in g["mess_dropout"]]
    right, _ = onp.ngcf_grads(w, adj, *ngcf_batch(g, 2), 1e-5, 24, ngcf_masks(g, 2), drop)
    assert_scalar_close(right, g["losses"][2], what="loss of the short batch")
    # dropout really is active in this fixture: without the masks the loss differs
    nomask, _ = onp.ngcf_grads(w, adj, *ngcf_batch(g, 2), 1e-5, 24)
    assert abs(nomask - right) > 1e-4


def test_reference_fp32_gradient_is_itself_ill_conditioned_on_small_activations():
    """Why the yardstick above is the fp64 evaluation: on the unscaled fixture the REFERENCE's own fp32
    gradient of the last hop is ~1e-4 of the tensor scale away from the exact one."""
    g = load_golden("ngcf_rmsprop_d64")
    w, adj = ngcf_params(g, "w0"), ngcf_adj(g)
    _, exact = onp.ngcf_grads(w, adj, *ngcf_batch(g, 0), 1e-5, 32, dt=np.float64)
    k = "Bi_weights.2.bias"
    rel_err = np.abs(g[f"g1/{k}"] - exact[k]).max() / np.abs(exact[k]).max()
    assert 2e-5 < rel_err < 1e-3
    k = "user_embedding.weight"
    assert np.abs(g[f"g1/{k}"] - exact[k]).max() / np.abs(exact[k]).max() < 2e-6


def test_predict_matches_reference():
    for case in CASES:
        g = load_golden(case)
        w = ngcf_params(g, f"w{int(g['meta'][5])}")
        scores = onp.ngcf_predict(w, ngcf_adj(g), g["probe_users"], g["probe_items"])
        assert_tensor_close(scores, g["probe_scores"], what=f"predict {case}")
