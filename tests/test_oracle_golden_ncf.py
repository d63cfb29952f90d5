"""Pin oracle/ncf_numpy.py (NeuMF / GMF / MLP restatement) against golden vectors captured from the
real reference (oracle/gen_golden.py --ncf).  CPU only."""
import numpy as np
import pytest

from helpers import assert_scalar_close, assert_tensor_close, load_golden
from oracle import ncf_numpy as onc

CASES = ["ncf_neumf_adam", "ncf_neumf_sgd_e32", "ncf_neumf_rmsprop_l2", "ncf_gmf_adam", "ncf_mlp_adam"]


def params(g, prefix):
    n = len(prefix) + 1
    return {k[n:]: g[k].astype(np.float32).copy() for k in g if k.startswith(prefix + "/")}


def opt_state(g, step, opt, like):
    st = onc.new_opt_state(like, opt)
    st["step"] = step
    if step > 0:
        if opt == "adam":
            st["exp_avg"], st["exp_avg_sq"] = params(g, f"m{step}"), params(g, f"v{step}")
        elif opt == "rmsprop":
            st["square_avg"] = params(g, f"v{step}")
    return st


@pytest.mark.parametrize("case", CASES)
def test_ncf_oracle_matches_reference(case):
    g = load_golden(case)
    n_steps = int(g["meta"][5])
    kind, opt, lr = str(g["kind"]), str(g["optimizer"]), float(g["lr"])
    for s in range(n_steps):
        w = params(g, f"w{s}")
        st = opt_state(g, s, opt, w)
        loss, grads, _ = onc.ncf_grads(w, g["users"][s], g["items"][s], g["ratings"][s], kind)
        assert_scalar_close(loss, g["losses"][s], what=f"loss step {s}")
        g_ref = params(g, f"g{s + 1}")
        assert set(grads) == set(g_ref)
        for k in g_ref:
            floor = 0.05 if k.endswith("bias") else 0.0  # sums of +/- terms (see helpers.py)
            assert_tensor_close(grads[k], g_ref[k], 2e-5, f"grad {k} step {s}", scale_floor=floor)
        if opt == "sgd":
            onc.opt_step(w, grads, st, opt, lr)
            for k in w:
                assert_tensor_close(w[k], g[f"w{s + 1}/{k}"], 1e-6, f"weights {k} step {s}")
    # predict contract: [n, 1] sigmoid scores of the final model
    wf = params(g, f"w{n_steps}")
    scores = onc.ncf_predict(wf, g["probe_users"], g["probe_items"], kind)
    assert scores.shape == g["probe_scores"].shape
    assert_tensor_close(scores, g["probe_scores"], 1e-5, "probe scores")


def test_neumf_init_quirk_q8_is_in_the_fixture():
    """init_weights never initialises embedding_item_mlp (stays N(0,1)); user tables get std 0.01."""
    g = load_golden("ncf_neumf_adam")
    assert 0.8 < g["w0/embedding_item_mlp.weight"].std() < 1.2
    assert g["w0/embedding_user_mlp.weight"].std() < 0.02
    assert g["w0/embedding_item_mf.weight"].std() < 0.02


def dropout_masks(g, s, w):
    """Keep masks of step s: one [B, in_features] boolean array per Linear of the tower."""
    B = int(g["meta"][4])
    out = []
    for l, li in enumerate(onc.layer_keys(w)):
        width = w[f"fc_layers.{li}.weight"].shape[1]
        out.append(np.unpackbits(g[f"mask{s}/{l}"], axis=1)[:, :width].astype(bool).reshape(B, width))
    return out


@pytest.mark.parametrize("case", ["ncf_neumf_dropout", "ncf_mlp_dropout"])
def test_ncf_oracle_with_tower_dropout_matches_reference(case):
    """Dropout > 0 in front of every Linear (ncf.py:42-45, mlp.py:30-33), masks captured from the
    reference's nn.Dropout modules (and shown by the generator to replay from the torch seed)."""
    g = load_golden(case)
    n_steps, p = int(g["meta"][5]), float(g["dropout"])
    kind, opt, lr = str(g["kind"]), str(g["optimizer"]), float(g["lr"])
    for s in range(n_steps):
        w = params(g, f"w{s}")
        st = opt_state(g, s, opt, w)
        masks = dropout_masks(g, s, w)
        assert all(abs(m.mean() - (1 - p)) < 0.1 for m in masks)
        loss, grads, _ = onc.ncf_grads(w, g["users"][s], g["items"][s], g["ratings"][s], kind, masks, p)
        assert_scalar_close(loss, g["losses"][s], what=f"loss step {s}")
        nodrop, _, _ = onc.ncf_grads(w, g["users"][s], g["items"][s], g["ratings"][s], kind)
        assert abs(nodrop - loss) > 1e-4, "the fixture is meant to exercise the masks"
        g_ref = params(g, f"g{s + 1}")
        for k in g_ref:
            floor = 0.05 if k.endswith("bias") else 0.0
            assert_tensor_close(grads[k], g_ref[k], 2e-5, f"grad {k} step {s}", scale_floor=floor)
        if opt == "sgd":
            onc.opt_step(w, grads, st, opt, lr)
            for k in w:
                assert_tensor_close(w[k], g[f"w{s + 1}/{k}"], 1e-6, f"weights {k} step {s}")
