"""GPU parity tests of the NCF family (NeuMF / GMF / MLP): libhiprec through the C ABI vs golden
vectors captured from the real reference and vs the numpy oracle at BASELINE's C3 shape.

Tolerances: 1e-5 relative (to the tensor's scale) on loss and gradients — the fp32 MFMA is an exact
fmaf chain, only the summation order differs from MKL's; optimizer updates within 1e-5 of the update
scale plus the Adam/RMSprop conditioning band (see tests/helpers.py::optimizer_band)."""
import contextlib
import io

import numpy as np
import pytest
import torch

from helpers import REL  # noqa: F401  (north_star: 1e-5 relative)
from helpers import EPS32, assert_grads_as_accurate, assert_scalar_close, assert_tensor_close, float64_oracle, load_golden, to64
from oracle import ncf_numpy as onc
from test_host_logic import ncf_config
from test_oracle_golden_ncf import opt_state, params

pytestmark = pytest.mark.gpu

CASES = [("ncf_neumf_adam", "NeuMFEngine"), ("ncf_neumf_sgd_e32", "NeuMFEngine"),
         ("ncf_neumf_rmsprop_l2", "NeuMFEngine"), ("ncf_gmf_adam", "GMFEngine"),
         ("ncf_mlp_adam", "MLPEngine")]


def bias_floor(k):
    return 0.05 if k.endswith("bias") else 0.0


def make_engine(engine, U, I, E, L, optimizer, lr, B):
    import beta_recsys_amd as hp

    with contextlib.redirect_stdout(io.StringIO()):
        return getattr(hp, engine)(ncf_config(U, I, E, L, optimizer, lr, B, device="cuda:0"))


def load_weights(eng, w):
    eng.model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})


def get_weights(eng):
    return {k: v.detach().cpu().numpy() for k, v in eng.model.state_dict().items()}


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("shape", [(64, 64, 32), (100, 37, 70), (4096, 128, 256), (1, 1, 1), (33, 129, 5)])
def test_gemm_f32_mfma_all_modes(hip_device, mode, shape):
    """Exact-fp32 MFMA GEMM vs numpy with ASYMMETRIC operands (catches transposed fragments)."""
    from beta_recsys_amd import _lib

    lib = _lib.load()
    M, N, K = shape
    rng = np.random.default_rng(M * 7 + N * 3 + K + mode)
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = rng.standard_normal((K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    mask = rng.standard_normal((M, N)).astype(np.float32)
    A = torch.from_numpy(a if mode != 2 else np.ascontiguousarray(a.T)).cuda()      # mode 2: [K, M]
    Bm = torch.from_numpy(np.ascontiguousarray(b.T) if mode == 0 else b).cuda()     # mode 0: [N, K]
    st = _lib.stream_ptr(hip_device)
    for use_bias, relu, use_mask in ((False, 0, False), (True, 1, False), (False, 0, True), (True, 1, True)):
        C = torch.full((M, N), float("nan"), device="cuda")
        mk = torch.from_numpy(mask).cuda()
        bs = torch.from_numpy(bias).cuda()
        _lib.check(lib.hiprec_gemm_f32(
            mode, M, N, K, _lib.ptr(A), A.shape[1], _lib.ptr(Bm), Bm.shape[1], _lib.ptr(C), N,
            _lib.ptr(bs) if use_bias else None, relu, _lib.ptr(mk) if use_mask else None, N, st))
        ref = a.astype(np.float64) @ b.astype(np.float64)
        if use_bias:
            ref = ref + bias
        if relu:
            ref = np.maximum(ref, 0)
        if use_mask:
            ref = ref * (mask > 0)
        scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64) + 1.0
        err = np.abs(C.cpu().numpy() - ref)
        assert np.all(err <= 4 * EPS32 * scale * np.sqrt(K) + 1e-6), (mode, shape, err.max())


@pytest.mark.parametrize("case,engine", CASES)
def test_ncf_step_matches_reference(hip_device, case, engine):
    g = load_golden(case)
    U, I, E, L, B, n_steps, _ = (int(x) for x in g["meta"])
    kind, opt, lr = str(g["kind"]), str(g["optimizer"]), float(g["lr"])
    eng = make_engine(engine, U, I, E, L, opt, lr, B)
    for s in range(n_steps):
        w_prev = params(g, f"w{s}")
        st_prev = opt_state(g, s, opt, w_prev)
        g_ref = params(g, f"g{s + 1}")
        batch = (torch.from_numpy(g["users"][s]), torch.from_numpy(g["items"][s]),
                 torch.from_numpy(g["ratings"][s]))
        load_weights(eng, w_prev)
        loss, grads = eng.backward_only(*batch)
        assert_scalar_close(loss, g["losses"][s], what=f"loss step {s}")
        with float64_oracle(onc):
            _, exact, _ = onc.ncf_grads(to64(w_prev), g["users"][s], g["items"][s], g["ratings"][s], kind)
        assert_grads_as_accurate(grads, g_ref, exact, f"grad step {s}", bias_floor)
        assert float(eng._g_flat.abs().max()) == 0.0
        # full step from the reference's own optimizer state
        eng.load_optimizer_state(s, st_prev.get("exp_avg"),
                                 st_prev.get("exp_avg_sq", st_prev.get("square_avg")))
        loss2 = eng.train_single_batch(*batch)
        assert_scalar_close(loss2, g["losses"][s], what=f"loss (step) {s}")
        # conditioning band: oracle step with the reference gradient perturbed by +/- 1e-5 * scale
        outs = []
        for sign in (1.0, -1.0):
            ww = {k: v.copy() for k, v in w_prev.items()}
            stc = {k: ({kk: vv.copy() for kk, vv in v.items()} if isinstance(v, dict) else v)
                   for k, v in st_prev.items()}
            gp = {k: (g_ref[k] + np.float32(sign * 1e-5 * max(np.abs(g_ref[k]).max(), bias_floor(k))))
                  for k in g_ref}
            onc.opt_step(ww, gp, stc, opt, lr)
            outs.append(ww)
        w = get_weights(eng)
        for k in w_prev:
            ref = g[f"w{s + 1}/{k}"].astype(np.float64)
            band = np.abs(outs[0][k].astype(np.float64) - outs[1][k])
            tol = 1e-5 * np.abs(ref - w_prev[k]).max() + 4 * EPS32 * np.abs(ref).max() + band
            err = np.abs(w[k] - ref)
            assert np.all(err <= tol), f"weights {k} step {s}: worst {err.max():.3e}"
        assert float(eng._g_flat.abs().max()) == 0.0
    # model.predict contract on the final reference weights
    load_weights(eng, params(g, f"w{n_steps}"))
    scores = eng.model.predict(g["probe_users"], g["probe_items"])
    assert isinstance(scores, torch.Tensor) and scores.shape == g["probe_scores"].shape
    got = scores.flatten().to(torch.device("cpu")).detach().numpy()
    assert_tensor_close(got, g["probe_scores"].reshape(-1), what="probe scores")


def test_ncf_trajectory_and_epoch(hip_device):
    """Three chained Adam steps through train_an_epoch: optimizer clock, prints, add_scalar."""
    g = load_golden("ncf_neumf_adam")
    U, I, E, L, B, n_steps, _ = (int(x) for x in g["meta"])
    eng = make_engine("NeuMFEngine", U, I, E, L, "adam", float(g["lr"]), B)
    load_weights(eng, params(g, "w0"))
    batches = [(torch.from_numpy(g["users"][s]), torch.from_numpy(g["items"][s]),
                torch.from_numpy(g["ratings"][s].astype(np.float64)))  # loader ratings may be double
               for s in range(n_steps)]
    with contextlib.redirect_stdout(io.StringIO()) as out:
        eng.train_an_epoch(batches, 0)
    assert "[Training Epoch 0], Loss" in out.getvalue()
    (tag, total, ep), = eng.writer.scalars
    assert tag == "model/loss" and ep == 0
    assert_scalar_close(total, float(np.sum(g["losses"])), REL, "epoch loss sum")
    # every element on the reference's trajectory: inside the envelope of oracle runs whose gradients are moved by
    # 1e-5 of their scale (helpers.oracle_trajectory), around the REAL engine's end point
    from helpers import assert_ncf_end_state

    np_batches = [(g["users"][s], g["items"][s], g["ratings"][s]) for s in range(n_steps)]
    assert_ncf_end_state(get_weights(eng), params(g, "w0"), np_batches, "neumf", "adam", float(g["lr"]),
                         "NeuMF trajectory", ref=params(g, f"w{n_steps}"))


def test_ncf_errors(hip_device):
    eng = make_engine("NeuMFEngine", 10, 10, 4, 2, "adam", 1e-3, 8)
    ok = (torch.tensor([1, 2, 3]), torch.tensor([3, 4, 5]), torch.tensor([1.0, 0.0, 1.0]))
    assert np.isfinite(eng.train_single_batch(*ok))
    with pytest.raises(IndexError):
        eng.train_single_batch(torch.tensor([1, 10]), torch.tensor([3, 4]), torch.tensor([1.0, 0.0]))
    assert np.isfinite(eng.train_single_batch(*ok))
    one = eng.train_single_batch(torch.tensor([1]), torch.tensor([2]), torch.tensor([1.0]))
    assert np.isfinite(one)  # unlike MF, a batch of one is legal for the NCF family


@pytest.mark.parametrize("E", [32, 64])
def test_ncf_full_size_c3_vs_oracle(hip_device, E):
    """BASELINE configs[2]: ML-1M shape, batch 4096; emb_dim 32 (tower 256->128->64->32) and emb_dim 64 -- "dim=64" as
    BASELINE.json spells it: tables 256/256/64/64, tower 512->256->128->64 -- both on the fused forward + chain launch
    (the wide tower's first layer in two 128-column passes, 134 KB of LDS)."""
    U, I, L, B = 6040, 3706, 3, 4096
    torch.manual_seed(5)
    eng = make_engine("NeuMFEngine", U, I, E, L, "adam", 1e-3, B)
    w = get_weights(eng)
    rng = np.random.default_rng(6)
    users, items = rng.integers(0, U, B), rng.integers(0, I, B)
    ratings = (rng.random(B) < 0.2).astype(np.float32)
    loss_ref, g_ref, _ = onc.ncf_grads(w, users, items, ratings, "neumf")
    loss, grads = eng.backward_only(torch.from_numpy(users), torch.from_numpy(items),
                                    torch.from_numpy(ratings))
    assert_scalar_close(loss, loss_ref, what="loss")
    with float64_oracle(onc):
        _, exact, _ = onc.ncf_grads(to64(w), users, items, ratings, "neumf")
    assert_grads_as_accurate(grads, g_ref, exact, "grad", bias_floor)
    scores = eng.model.predict(users[:1000], items[:1000]).cpu().numpy()
    assert_tensor_close(scores, onc.ncf_predict(w, users[:1000], items[:1000], "neumf"), what="scores")


@pytest.mark.parametrize("engine,kind,E,L,B", [("MLPEngine", "mlp", 32, 3, 1000), ("NeuMFEngine", "neumf", 16, 2, 77),
                                                ("MLPEngine", "mlp", 64, 2, 333), ("NeuMFEngine", "neumf", 64, 3, 100),
                                                ("NeuMFEngine", "neumf", 16, 2, 40000)])
def test_fused_tower_shapes_vs_oracle(hip_device, engine, kind, E, L, B):
    """Shapes that take the fused launch (2*dim_mlp <= 512, layer widths <= 256, multiples of 32) in the
    variants the goldens do not reach: the stand-alone MLP (no ReLU on the embeddings, no GMF half),
    two-layer towers, ragged last blocks, the 512-wide input with a 256-wide first layer (two output passes) on a
    ragged batch, and a batch beyond the fused forward's 32 768 samples (launch-per-layer forward, then the
    input-gradient chain in its own launch); gradients and scores vs the numpy oracle."""
    U, I = 700, 500
    torch.manual_seed(E + L)
    eng = make_engine(engine, U, I, E, L, "adam", 1e-3, B)
    w = get_weights(eng)
    rng = np.random.default_rng(B)
    users, items = rng.integers(0, U, B), rng.integers(0, I, B)
    ratings = (rng.random(B) < 0.3).astype(np.float32)
    loss_ref, g_ref, _ = onc.ncf_grads(w, users, items, ratings, kind)
    loss, grads = eng.backward_only(torch.from_numpy(users), torch.from_numpy(items), torch.from_numpy(ratings))
    assert_scalar_close(loss, loss_ref, what="loss")
    with float64_oracle(onc):
        _, exact, _ = onc.ncf_grads(to64(w), users, items, ratings, kind)
    assert_grads_as_accurate(grads, g_ref, exact, "grad", bias_floor)
    scores = eng.model.predict(users[:200], items[:200]).cpu().numpy()
    assert_tensor_close(scores, onc.ncf_predict(w, users[:200], items[:200], kind), what="scores")


@pytest.mark.parametrize("case,engine", [("ncf_neumf_dropout", "NeuMFEngine"), ("ncf_mlp_dropout", "MLPEngine")])
def test_tower_dropout_matches_reference(hip_device, case, engine):
    """Dropout > 0 in front of every Linear of the tower (ncf.py:42-45, mlp.py:30-33): with the reference's
    torch seed the engine draws the SAME masks (CPU replay of nn.Dropout), and loss, gradients and the SGD
    step match the real reference."""
    from test_oracle_golden_ncf import dropout_masks

    g = load_golden(case)
    U, I, E, L, B, n_steps, seed = (int(x) for x in g["meta"])
    kind, opt, lr, p = str(g["kind"]), str(g["optimizer"]), float(g["lr"]), float(g["dropout"])
    import beta_recsys_amd as hp

    cfg = ncf_config(U, I, E, L, opt, lr, B, device="cuda:0")
    cfg["model"]["dropout"] = p
    with contextlib.redirect_stdout(io.StringIO()):
        eng = getattr(hp, engine)(cfg)
    eng.model.train()
    for s in range(n_steps):
        w0 = params(g, f"w{s}")
        load_weights(eng, w0)
        eng.load_optimizer_state(s)      # gradients / the SGD step do not depend on the moments
        batch = (g["users"][s], g["items"][s], g["ratings"][s])
        torch.manual_seed(3000 + s)
        loss, grads = eng.backward_only(*batch)
        ref_masks = dropout_masks(g, s, w0)
        for l, ref in enumerate(ref_masks):
            mine = eng.model._ws["keep"][l][:B].cpu().numpy().astype(bool)
            assert np.array_equal(mine, ref), f"layer {l}: the same seed must drop the same activations"
        assert_scalar_close(loss, g["losses"][s], what=f"loss step {s}")
        g_ref = params(g, f"g{s + 1}")
        with float64_oracle(onc):
            _, exact, _ = onc.ncf_grads(to64(w0), *batch, kind, masks=ref_masks, dropout=p)
        assert_grads_as_accurate(grads, g_ref, exact, f"grad step {s}", bias_floor)
        if opt == "sgd":
            torch.manual_seed(3000 + s)
            loss2 = eng.train_single_batch(*(torch.from_numpy(x) for x in batch))
            assert_scalar_close(loss2, g["losses"][s], what=f"loss (step) {s}")
            w1 = get_weights(eng)
            for k in w0:
                assert_tensor_close(w1[k], g[f"w{s + 1}/{k}"], 1e-6, f"weights {k} step {s}")
    # eval mode: no dropout, the fused path scores like the oracle without masks
    eng.model.eval()
    wf = params(g, f"w{n_steps}")
    load_weights(eng, wf)
    probe_u, probe_i = g["users"][0][:16], g["items"][0][:16]
    scores = eng.model.predict(probe_u, probe_i).cpu().numpy()
    assert_tensor_close(scores, onc.ncf_predict(wf, probe_u, probe_i, kind), 1e-5, "eval scores")
    # device RNG: keep rate ~ 1 - p and a different mask every step
    cfg["model"]["dropout_rng"] = "device"
    with contextlib.redirect_stdout(io.StringIO()):
        dev_eng = getattr(hp, engine)(cfg)
    dev_eng.model.train()
    dev_eng.train_single_batch(*(torch.from_numpy(x) for x in batch))
    k1 = dev_eng.model._ws["keep"][0][:B].clone()
    dev_eng.train_single_batch(*(torch.from_numpy(x) for x in batch))
    assert abs(float(k1.float().mean()) - (1 - p)) < 0.12 and not torch.equal(k1, dev_eng.model._ws["keep"][0][:B])


@pytest.mark.parametrize("kind,engine,E,p", [("neumf", "NeuMFEngine", 32, 0.3), ("mlp", "MLPEngine", 32, 0.5),
                                             ("neumf", "NeuMFEngine", 64, 0.2)])
def test_tower_dropout_inside_the_fused_forward(hip_device, kind, engine, E, p):
    """Tower dropout at a shape the FUSED tower kernel takes (emb_dim 32: 256 -> 128 -> 64 -> 32, batch 4096 + a
    ragged tail; emb_dim 64: the 512-wide input, two output passes in layer 1 and four in the chain's last layer, the
    ninth wave copying tiles of every width): the keep bytes of every Linear's input are applied inside the fused
    launch (gather for layer 0, layer epilogues after that).  Loss, every gradient and the scores' keep rate against oracle/ncf_numpy.py fed
    with the very masks the engine drew (read back from its workspace)."""
    import beta_recsys_amd as hp

    U, I, L, B = 6040, 3706, 3, 4096 + 37
    cfg = ncf_config(U, I, E, L, "sgd", 0.05, B, device="cuda:0")
    cfg["model"]["dropout"] = p
    cfg["model"]["dropout_rng"] = "device"
    torch.manual_seed(11)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = getattr(hp, engine)(cfg)
    eng.model.train()
    rng = np.random.default_rng(2)
    users, items = rng.integers(0, U, B), rng.integers(0, I, B)
    ratings = (rng.random(B) < 0.2).astype(np.float32)
    w0 = get_weights(eng)
    loss, grads = eng.backward_only(users, items, ratings)
    n_layers = len(eng.model.tower_dims)
    masks = [eng.model._ws["keep"][l][:B].cpu().numpy().astype(bool) for l in range(n_layers)]
    for m in masks:
        assert abs(m.mean() - (1 - p)) < 0.02
    ref_loss, g_ref, _ = onc.ncf_grads(w0, users, items, ratings, kind, masks=masks, dropout=p)
    assert_scalar_close(loss, ref_loss, what="loss with tower dropout (fused forward)")
    with float64_oracle(onc):
        _, exact, _ = onc.ncf_grads(to64(w0), users, items, ratings, kind, masks=masks, dropout=p)
    assert_grads_as_accurate(grads, g_ref, exact, "grad", bias_floor)


@pytest.mark.parametrize("mode", ["split", "unfused"])
def test_older_backward_forms_stay_correct(hip_device, mode):
    """hiprec_ncf_grad's default is ONE launch for forward + head + input-gradient chain + scatter.  The two older
    forms it falls back to -- the chain in a launch of its own (forward on the launch-per-layer path: batch beyond
    32 768), one grouped launch per layer + scatter (shapes outside the fused limits) -- are selected here through
    HIPREC_NCF_BACKWARD, which only libhiprec_test.so (-DHIPREC_TEST_SWITCHES) reads (once per process: a fresh
    interpreter), and run the step / dropout / full-size tests."""
    import os
    import subprocess
    import sys

    if os.environ.get("HIPREC_NCF_BACKWARD"):
        pytest.skip("already running an older form")
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.join(here, "test_ncf_gpu.py"), "-m", "gpu", "-x", "-q", "-k",
         "not older_backward_forms"],
        env=dict(os.environ, HIPREC_NCF_BACKWARD=mode, HIPREC_LIB="libhiprec_test.so"), capture_output=True, text=True,
        timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout
