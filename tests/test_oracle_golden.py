"""Pin the oracle (oracle/mf_numpy.py, oracle/torch_port.py) against golden vectors captured from
the real reference by oracle/gen_golden.py.  CPU only."""
import numpy as np
import pytest
import torch

from helpers import KEYS, assert_scalar_close, assert_tensor_close, assert_update_close
from helpers import assert_step_close, golden_opt_state, grad_scale_floor, load_golden
from helpers import optimizer_band, params
from oracle import mf_numpy as onp
from oracle.torch_port import TorchMFPort

STEP_CASES = ["mf_bpr_sgd", "mf_bpr_adam", "mf_bpr_rmsprop", "mf_bce_sgd", "mf_bce_adam",
              "mf_bpr_sgd_d4", "mf_bpr_sgd_d100", "mf_bpr_adam_d200", "mf_bpr_sgd_d300",
              "mf_bpr_sgd_reg", "mf_bce_sgd_reg",
              # one item holding 60 % of a 320-triple batch's positives, captured from the real engine
              "mf_bpr_adam_hot", "mf_bpr_sgd_hot"]


def test_known_answer_vector():
    """SURVEY.md §8c known-answer: loss 0.685829461, regularizer 0.818666756."""
    g = load_golden("mf_kat")
    assert_scalar_close(float(g["loss"]), 0.685829461, 1e-6, "golden loss vs survey")
    assert_scalar_close(float(g["reg"]), 0.818666756, 1e-6, "golden reg vs survey")
    w = params(g, "w0")
    st = onp.new_opt_state(w, "sgd")
    loss, reg = onp.mf_train_step(w, st, (g["users"], g["pos"], g["neg"]), "bpr", "sgd", 0.1)
    assert_scalar_close(loss, float(g["loss"]), what="loss")
    assert_scalar_close(reg, float(g["reg"]), what="reg")
    for k in KEYS:
        assert_update_close(g[f"w0/{k}"], w[k], g[f"w1/{k}"], what=k)
    # SGD leaves untouched rows bit-identical (rows 1 and 3 of user_emb)
    assert np.array_equal(w["user_emb.weight"][[1, 3]], g["w0/user_emb.weight"][[1, 3]])


@pytest.mark.parametrize("case", STEP_CASES)
def test_numpy_oracle_matches_reference(case):
    """Every step in isolation, started from the reference's own weights and optimizer state."""
    g = load_golden(case)
    B, n_steps = int(g["meta"][3]), int(g["meta"][4])
    opt, loss_kind, lr = str(g["optimizer"]), str(g["loss_kind"]), float(g["lr"])
    for s in range(n_steps):
        w = params(g, f"w{s}")
        st = golden_opt_state(g, s, opt)
        batch = (g["users"][s], g["items_a"][s], g["third"][s])
        reg_coef = float(g["reg_coef"])
        if loss_kind == "bpr":
            lval, reg, grads = onp.mf_bpr_grads(w, *batch, reg_coef)
        else:
            lval, reg, grads = onp.mf_bce_grads(w, *batch, reg_coef)
        assert_scalar_close(lval, g["losses"][s], what=f"loss step {s}")
        assert_scalar_close(reg, g["regs"][s], what=f"reg step {s}")
        g_ref = params(g, f"g{s + 1}")
        for k in KEYS:
            assert_tensor_close(grads[k], g_ref[k], what=f"grad {k} step {s}",
                                scale_floor=grad_scale_floor(k, B))
        band = optimizer_band(w, st, g_ref, opt, lr, B)
        w_prev = onp.copy_params(w)
        onp.opt_step(w, grads, st, opt, lr)
        for k in KEYS:
            assert_step_close(w_prev[k], w[k], g[f"w{s + 1}/{k}"], band[k],
                              what=f"weights {k} step {s}")
        ref_next = golden_opt_state(g, s + 1, opt)
        for name in ("exp_avg", "exp_avg_sq", "square_avg"):
            if name in st:
                for k in KEYS:
                    floor = grad_scale_floor(k, B)
                    assert_tensor_close(st[name][k], ref_next[name][k], 4e-5,
                                        f"{name} {k} step {s}",
                                        scale_floor=floor if name == "exp_avg" else floor ** 2)


def test_numpy_oracle_multi_step_trajectory():
    """All steps chained (state carried by the oracle itself): catches step-counter / bias-
    correction mistakes that per-step isolation cannot."""
    for case in ("mf_bpr_adam", "mf_bpr_rmsprop", "mf_bpr_sgd"):
        g = load_golden(case)
        n_steps = int(g["meta"][4])
        opt, lr = str(g["optimizer"]), float(g["lr"])
        w = params(g, "w0")
        st = onp.new_opt_state(w, opt)
        for s in range(n_steps):
            lval, _ = onp.mf_train_step(w, st, (g["users"][s], g["items_a"][s], g["third"][s]),
                                        "bpr", opt, lr)
            assert_scalar_close(lval, g["losses"][s], 2e-5, f"{case} loss step {s}")
        for k in KEYS:  # loose: ill-conditioned Adam elements drift (see helpers.optimizer_band)
            frac_bad = np.mean(np.abs(w[k] - g[f"w{n_steps}/{k}"]) > 1e-3 * lr + 1e-6)
            assert frac_bad < 0.01, f"{case} {k}: {frac_bad:.3%} of elements off trajectory"


@pytest.mark.parametrize("case", STEP_CASES)
def test_torch_port_matches_reference(case):
    g = load_golden(case)
    n_steps = int(g["meta"][4])
    opt, loss_kind, lr = str(g["optimizer"]), str(g["loss_kind"]), float(g["lr"])
    port = TorchMFPort(params(g, "w0"), opt, lr, loss_kind, float(g["reg_coef"]))
    for s in range(n_steps):
        third = torch.from_numpy(g["third"][s])
        batch = (torch.from_numpy(g["users"][s]), torch.from_numpy(g["items_a"][s]), third)
        lval, reg = port.step(batch)
        # same ATen ops as the reference -> expected to be (near) bit-identical
        assert_scalar_close(lval, g["losses"][s], 1e-6, f"loss step {s}")
        assert_scalar_close(reg, g["regs"][s], 1e-6, f"reg step {s}")
        w = port.numpy_weights()
        for k in KEYS:
            assert_tensor_close(w[k], g[f"w{s + 1}/{k}"], 1e-6, f"weights {k} step {s}")


def test_predict_oracle():
    g = load_golden("mf_predict")
    w = {k: g[f"w/{k}"] for k in KEYS}
    assert_tensor_close(onp.mf_predict(w, g["users"], g["items"]), g["scores"], what="scores")


def test_dense_adam_moves_untouched_rows():
    """SURVEY §7 hard part: with dense Adam a row touched only at step 1 keeps moving."""
    g = load_golden("mf_bpr_adam")
    touched1 = set(g["users"][0].tolist())
    later = set(g["users"][1].tolist()) | set(g["users"][2].tolist())
    only_first = sorted(touched1 - later)
    assert only_first, "fixture should contain a user row touched at step 1 only"
    r = only_first[0]
    assert not np.array_equal(g["w1/user_emb.weight"][r], g["w2/user_emb.weight"][r])
    assert not np.array_equal(g["w2/user_emb.weight"][r], g["w3/user_emb.weight"][r])


def test_c1_config_losses_and_checksums_from_the_reference():
    """BASELINE configs[0] (configs/mf_default.json shape: 943 x 1682, emb_dim 64, batch 400, adam lr 0.05)
    run by the REAL reference from its own seeded init: the oracle, started from the mirror's init for
    the same torch seed, reproduces its losses and weight checksums."""
    import contextlib
    import io

    import torch

    import beta_recsys_amd as hp
    from oracle import mf_numpy as onp

    g = load_golden("mf_c1_adam")
    U, I, D, B, n_steps, seed = (int(x) for x in g["meta"])
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        model = hp.MF(dict(n_users=U, n_items=I, emb_dim=D, device_str="cpu"))
    w = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    st = onp.new_opt_state(w, "adam")
    for s in range(n_steps):
        loss, reg = onp.mf_train_step(w, st, (g["users"][s], g["pos"][s], g["neg"][s]), "bpr", "adam", 0.05)
        assert_scalar_close(loss, g["losses"][s], 1e-5, f"loss of step {s}")
        assert_scalar_close(reg, g["regs"][s], 1e-5, f"regularizer of step {s}")
    for k in KEYS:
        a = w[k].astype(np.float64)
        assert abs((a * a).sum() - float(g[f"sumsq/{k}"])) <= 2e-3 * float(g[f"sumsq/{k}"]) + 1e-9, k
        head = w[k].reshape(-1)[:64]
        close = np.abs(head - g[f"head/{k}"]) <= 2e-3 * max(np.abs(g[f"head/{k}"]).max(), 1e-3)
        assert close.mean() >= 0.9, f"{k}: {close.mean():.2f} of the sampled weights agree"
