"""Pin oracle/triple2vec_numpy.py against golden vectors captured from the real reference's
Triple2vecEngine and AliasTable by ``oracle/gen_golden.py --t2v``.  CPU only."""
import numpy as np
import pytest

from helpers import REL, assert_scalar_close, assert_step_close, assert_tensor_close, copy_state, load_golden
from oracle import triple2vec_numpy as onp

CASES = ["t2v_adam", "t2v_sgd_d100", "t2v_rmsprop_init"]
KEYS = onp.KEYS
BATCH_KEYS = ("pos_u", "pos_i1", "pos_i2", "neg_u", "neg_i1", "neg_i2")


def t2v_params(g, prefix):
    return {k: g[f"{prefix}/{k}"].astype(np.float32).copy() for k in KEYS}


def t2v_batch(g, s):
    return tuple(g[f"b{s}/{k}"] for k in BATCH_KEYS)


def t2v_opt_state(g, step, opt):
    st = onp.new_opt_state(t2v_params(g, "w0"), opt)
    st["step"] = step
    if step > 0 and opt == "adam":
        st["exp_avg"], st["exp_avg_sq"] = t2v_params(g, f"m{step}"), t2v_params(g, f"v{step}")
    elif step > 0 and opt == "rmsprop":
        st["square_avg"] = t2v_params(g, f"v{step}")
    return st


def bias_floor(key, scale):
    """A bias gradient sums terms of size <= scale with both signs (see helpers.grad_scale_floor)."""
    return scale if key.endswith("bias.weight") else 0.0


def t2v_band(w_prev, st_prev, g_ref, opt, lr, scale, rel=REL):
    """Forward-error band of one optimizer step for a gradient within rel * its scale of g_ref."""
    outs = []
    for sign in (+1.0, -1.0):
        w = {k: v.copy() for k, v in w_prev.items()}
        st = copy_state(st_prev)
        g = {k: (g_ref[k] + np.float32(sign * rel * max(float(np.abs(g_ref[k]).max()), bias_floor(k, scale))))
             .astype(np.float32) for k in KEYS}
        g["item_emb2.weight"] = g_ref["item_emb2.weight"].copy()   # the orphaned table has no gradient at all
        onp.opt_step(w, g, st, opt, lr)
        outs.append(w)
    return {k: np.abs(outs[0][k].astype(np.float64) - outs[1][k].astype(np.float64)) for k in KEYS}


@pytest.mark.parametrize("case", CASES)
def test_numpy_oracle_matches_reference(case):
    """Every step in isolation from the reference's own weights and optimizer state."""
    g = load_golden(case)
    B, n_steps = int(g["meta"][3]), int(g["meta"][5])
    opt, lr = str(g["optimizer"]), float(g["lr"])
    scale = 1.0 / (3 * B)
    for s in range(n_steps):
        w = t2v_params(g, f"w{s}")
        st = t2v_opt_state(g, s, opt)
        loss, grads = onp.t2v_grads(w, t2v_batch(g, s), B)
        assert_scalar_close(loss, g["losses"][s], what=f"loss step {s}")
        g_ref = t2v_params(g, f"g{s + 1}")
        assert not g_ref["item_emb2.weight"].any(), "item_emb2 is aliased away: its own parameter has no grad"
        for k in KEYS:
            assert_tensor_close(grads[k], g_ref[k], what=f"grad {k} step {s}", scale_floor=bias_floor(k, scale))
        band = t2v_band(w, st, g_ref, opt, lr, scale)
        w_prev = {k: v.copy() for k, v in w.items()}
        onp.opt_step(w, grads, st, opt, lr)
        w["item_emb2.weight"][...] = w["item_emb1.weight"]
        band["item_emb2.weight"] = band["item_emb1.weight"]
        w_prev["item_emb2.weight"] = w_prev["item_emb1.weight"]
        for k in KEYS:
            assert_step_close(w_prev[k], w[k], g[f"w{s + 1}/{k}"], band[k], what=f"weights {k} step {s}")
        nxt = t2v_opt_state(g, s + 1, opt)
        for name in ("exp_avg", "exp_avg_sq", "square_avg"):
            if name in st:
                for k in KEYS:
                    floor = bias_floor(k, scale)
                    assert_tensor_close(st[name][k], nxt[name][k], 4e-5, f"{name} {k} step {s}",
                                        scale_floor=floor if name == "exp_avg" else floor ** 2)


def test_short_batch_still_divides_by_configured_batch_size():
    """triple2vec.py:92: the last (short) batch of t2v_adam has 5 triples but the loss is / (3 * 12)."""
    g = load_golden("t2v_adam")
    assert list(g["batch_lens"]) == [12, 12, 5]
    w = t2v_params(g, "w2")
    right, _ = onp.t2v_grads(w, t2v_batch(g, 2), 12)
    wrong, _ = onp.t2v_grads(w, t2v_batch(g, 2), 5)
    assert_scalar_close(right, g["losses"][2], what="loss")
    assert abs(wrong - g["losses"][2]) > 0.5


def test_item_rows_follow_neg_i_2_and_bias_follows_neg_i_1():
    """triple2vec.py:46,69-71: swapping neg_i_1 for other ids with the same bias leaves the loss
    unchanged, changing neg_i_2 does not."""
    g = load_golden("t2v_adam")
    w = t2v_params(g, "w0")
    w["item_bias.weight"][...] = 0.25
    b = list(t2v_batch(g, 0))
    base, _ = onp.t2v_grads(w, b, 12)
    b1 = list(b)
    b1[4] = (b[4] + 1) % 17
    assert onp.t2v_grads(w, b1, 12)[0] == base
    b2 = list(b)
    b2[5] = (b[5] + 1) % 17
    assert onp.t2v_grads(w, b2, 12)[0] != base


def test_trajectory_and_predict():
    g = load_golden("t2v_sgd_d100")
    B, n_steps = int(g["meta"][3]), int(g["meta"][5])
    w = t2v_params(g, "w0")
    st = onp.new_opt_state(w, "sgd")
    for s in range(n_steps):
        loss = onp.t2v_train_step(w, st, t2v_batch(g, s), B, "sgd", float(g["lr"]))
        assert_scalar_close(loss, g["losses"][s], 5e-5, what=f"loss step {s}")
    for k in KEYS:
        assert_tensor_close(w[k], g[f"w{n_steps}/{k}"], 5e-5, what=f"final {k}")
    for case in CASES:
        g = load_golden(case)
        w = t2v_params(g, f"w{int(g['meta'][5])}")
        assert_tensor_close(onp.t2v_predict(w, g["predict/users"], g["predict/items"]), g["predict/scores"],
                            what=f"predict {case}")


def test_alias_table_construction_matches_reference():
    g = load_golden("alias_table")
    for tag in ("zipf", "flat", "rand"):
        prob, alias = onp.build_alias_table(g[f"{tag}/freq"].tolist())
        assert np.array_equal(alias, g[f"{tag}/alias"]), tag
        assert np.array_equal(prob, g[f"{tag}/prob"]), tag


def test_alias_sample_restatement_follows_the_frequencies():
    g = load_golden("alias_table")
    freq = g["zipf/freq"].astype(np.float64)
    n = 400_000
    s = onp.alias_sample(g["zipf/prob"], g["zipf/alias"], None, 99, n)
    assert s.min() >= 0 and s.max() < len(freq)
    expected = n * freq / freq.sum()
    got = np.bincount(s, minlength=len(freq))
    assert np.all(np.abs(got - expected) < 5 * np.sqrt(expected) + 5)
    labels = np.arange(len(freq))[::-1] * 10
    assert np.array_equal(onp.alias_sample(g["zipf/prob"], g["zipf/alias"], labels, 99, 1000), labels[s[:1000]])
    assert not np.array_equal(s[:1000], onp.alias_sample(g["zipf/prob"], g["zipf/alias"], None, 100, 1000))
