"""GPU parity tests of the MF hot path: libhiprec (HIP, through the C ABI) vs the golden vectors
captured from the real reference, vs the numpy oracle on seeded inputs, and size-independent
properties at BASELINE's full C2 shape (6040 x 3706, dim 64, batch 4096).

Tolerances (BASELINE.json north_star): bit-exact on the index gather; 1e-5 relative on fp32
loss / gradients (relative to the tensor's scale, see helpers.py); optimizer updates within 1e-5
of the update scale plus the conditioning band of helpers.optimizer_band.
"""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from helpers import EPS32, KEYS, REL, assert_scalar_close, assert_step_close, assert_tensor_close, assert_update_close
from helpers import assert_mf_end_state, golden_opt_state, grad_scale_floor, legal_trajectory_envelope, load_golden
from helpers import assert_on_trajectory, check_mf_end_state, mf_trajectory, optimizer_band, params
from oracle import mf_numpy as onp

pytestmark = pytest.mark.gpu

STEP_CASES = ["mf_bpr_sgd", "mf_bpr_adam", "mf_bpr_rmsprop", "mf_bce_sgd", "mf_bce_adam",
              "mf_bpr_sgd_d4", "mf_bpr_sgd_d100", "mf_bpr_adam_d200", "mf_bpr_sgd_d300",
              "mf_bpr_sgd_reg", "mf_bce_sgd_reg",
              # one item holding 60 % of a 320-triple batch's positives, captured from the real engine
              "mf_bpr_adam_hot", "mf_bpr_sgd_hot"]


def make_engine(U, I, D, optimizer, loss, lr, B, reg=None, **model_extra):
    import beta_recsys_amd as hp

    model = dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer=optimizer, lr=lr,
                 batch_size=B, loss=loss)
    model.update(model_extra)
    cfg = {"model": model, "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    if reg is not None:
        cfg["reg"] = reg
        cfg["model"]["reg"] = reg
    with contextlib.redirect_stdout(io.StringIO()):
        eng = hp.MFEngine(cfg)
    return eng


def load_weights(eng, w):
    eng.model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})


def get_weights(eng):
    return {k: v.detach().cpu().numpy() for k, v in eng.model.state_dict().items()}


def engine_for_case(g, **extra):
    U, I, D, B, _ = (int(x) for x in g["meta"])
    reg = float(g["reg_coef"]) or None
    return make_engine(U, I, D, str(g["optimizer"]), str(g["loss_kind"]), float(g["lr"]), B,
                       reg=reg, **extra)


def batch_of(g, s, device="cpu"):
    return tuple(torch.from_numpy(g[k][s]).to(device) for k in ("users", "items_a", "third"))


# ---- gather: integer-indexed copy must be bit-exact ---------------------------------------------

@pytest.mark.parametrize("dim", [1, 4, 63, 64, 100, 128, 256])
def test_gather_rows_bit_exact(hip_device, dim):
    import beta_recsys_amd as hp

    gen = torch.Generator().manual_seed(dim)
    table = torch.randn(1000, dim, generator=gen)
    idx = torch.randint(0, 1000, (4097,), generator=gen)
    idx[:10] = idx[0]  # duplicates
    out = hp.gather_rows(table.to(hip_device), idx.to(hip_device))
    assert torch.equal(out.cpu(), table[idx]), "gather must be a bit-exact copy"
    out2 = hp.gather_rows(table.to(hip_device), idx[:0].to(hip_device))
    assert out2.shape == (0, dim)


def test_gather_rows_2d_index_and_oob(hip_device):
    import beta_recsys_amd as hp
    from beta_recsys_amd import mf as hmf

    table = torch.arange(40, dtype=torch.float32).reshape(10, 4).to(hip_device)
    idx = torch.tensor([[0, 9], [3, 3]])
    out = hp.gather_rows(table, idx)
    assert out.shape == (2, 2, 4)
    assert torch.equal(out.cpu(), table.cpu()[idx])
    oob = hp.gather_rows(table, torch.tensor([1, 10]))  # 10 is out of range: zero row + status bit
    assert torch.equal(oob.cpu(), torch.stack([table.cpu()[1], torch.zeros(4)]))
    key = (hip_device.type, hip_device.index)
    st = hmf.read_stats(hmf._gather_stats[key])
    assert st.status & hp._lib.STATUS_ROW_OOB
    hmf._gather_stats.pop(key)
    with pytest.raises(IndexError):
        hp.gather_rows(table, torch.tensor([1, 10]), check=True)
    assert torch.equal(hp.gather_rows(table, torch.tensor([2]), check=True).cpu(), table.cpu()[[2]])


# ---- golden vectors from the real reference -----------------------------------------------------

def test_known_answer_vector(hip_device):
    g = load_golden("mf_kat")
    eng = make_engine(4, 5, 4, "sgd", "bpr", 0.1, 3)
    load_weights(eng, params(g, "w0"))
    loss, reg = eng.train_single_batch(tuple(torch.from_numpy(g[k]) for k in ("users", "pos", "neg")))
    assert_scalar_close(loss, float(g["loss"]), what="loss")
    assert_scalar_close(reg, float(g["reg"]), what="reg")
    w = get_weights(eng)
    for k in KEYS:
        assert_step_close(g[f"w0/{k}"], w[k], g[f"w1/{k}"], 0.0, what=k)
    assert np.array_equal(w["user_emb.weight"][[1, 3]], g["w0/user_emb.weight"][[1, 3]])


@pytest.mark.parametrize("case", STEP_CASES)
def test_step_matches_reference(hip_device, case):
    """Each step in isolation from the reference's own weights + optimizer state: loss, reg,
    dense gradients, updated weights and updated moments."""
    g = load_golden(case)
    B, n_steps = int(g["meta"][3]), int(g["meta"][4])
    opt, lr = str(g["optimizer"]), float(g["lr"])
    eng = engine_for_case(g)
    for s in range(n_steps):
        w_prev = params(g, f"w{s}")
        st_prev = golden_opt_state(g, s, opt)
        g_ref = params(g, f"g{s + 1}")
        # (1) gradients as autograd leaves them
        load_weights(eng, w_prev)
        loss, reg, grads = eng.backward_only(batch_of(g, s))
        assert_scalar_close(loss, g["losses"][s], what=f"loss step {s}")
        assert_scalar_close(reg, g["regs"][s], what=f"reg step {s}")
        for k in KEYS:
            assert_tensor_close(grads[k].cpu().numpy(), g_ref[k], what=f"grad {k} step {s}",
                                scale_floor=grad_scale_floor(k, B))
        assert float(eng._g_flat.abs().max()) == 0.0
        # (2) the full step
        eng.load_optimizer_state(s, st_prev.get("exp_avg"),
                                 st_prev.get("exp_avg_sq", st_prev.get("square_avg")))
        loss2, reg2 = eng.train_single_batch(batch_of(g, s))
        assert_scalar_close(loss2, g["losses"][s], what=f"loss (step) {s}")
        assert_scalar_close(reg2, g["regs"][s], what=f"reg (step) {s}")
        band = optimizer_band(w_prev, st_prev, g_ref, opt, lr, B)
        w = get_weights(eng)
        for k in KEYS:
            assert_step_close(w_prev[k], w[k], g[f"w{s + 1}/{k}"], band[k],
                              what=f"weights {k} step {s}")
        assert float(eng._g_flat.abs().max()) == 0.0, "optimizer must leave the grad buffer zeroed"
        step, m, v = eng.optimizer_state()
        assert step == s + 1
        ref_next = golden_opt_state(g, s + 1, opt)
        for name, got in (("exp_avg", m), ("exp_avg_sq", v), ("square_avg", v)):
            if name in ref_next and got is not None:
                for k in KEYS:
                    floor = grad_scale_floor(k, B)
                    # exp_avg is linear in the gradient (1e-5), the second moments quadratic (2e-5)
                    assert_tensor_close(got[k].cpu().numpy(), ref_next[name][k], REL if name == "exp_avg" else 2 * REL,
                                        f"{name} {k} step {s}",
                                        scale_floor=floor if name == "exp_avg" else floor ** 2)


IEEE_BUILD = os.environ.get("HIPREC_LIB", "").endswith("ieee.so")


@pytest.mark.parametrize("case", ["mf_bpr_sgd", "mf_bpr_adam", "mf_bpr_rmsprop", "mf_bce_adam", "mf_bpr_adam_hot"])
def test_multi_step_trajectory(hip_device, case):
    """All steps chained on the GPU (its own state): optimizer clock and bias correction.

    The bound is elementwise and derived, with no allowance for a fraction of outliers: north_star lets every
    gradient be off by 1e-5 of its tensor's scale, and helpers.legal_trajectory_envelope runs the oracle with such
    perturbations to see how far a legal n-step trajectory can end up from the reference's, element by element
    (Adam / RMSprop turn a 1e-5 gradient error into up to ~lr where the gradient is ~eps and their moments carry it
    on, which is why the old test tolerated 1 % of elements).  Every element must lie within twice that envelope plus
    1e-5 of the largest update and 4 ulp of the weights -- with the default build (v_rcp_f32 / v_sqrt_f32 in the
    denominators) AND with libhiprec_ieee.so (ATen's correctly rounded sqrt / division)."""
    g = load_golden(case)
    B, n_steps, lr, opt = int(g["meta"][3]), int(g["meta"][4]), float(g["lr"]), str(g["optimizer"])
    eng = engine_for_case(g)
    load_weights(eng, params(g, "w0"))
    for s in range(n_steps):
        loss, reg = eng.train_single_batch(batch_of(g, s, "cuda:0"))
        assert_scalar_close(loss, g["losses"][s], REL, f"loss step {s}")
        assert_scalar_close(reg, g["regs"][s], REL, f"reg step {s}")
    w = get_weights(eng)
    env = legal_trajectory_envelope(g, n_steps, opt, lr, B, str(g["loss_kind"]), float(g["reg_coef"]))
    for k in KEYS:
        ref = g[f"w{n_steps}/{k}"].astype(np.float64)
        upd = max(np.abs(g[f"w{s + 1}/{k}"].astype(np.float64) - g[f"w{s}/{k}"]).max() for s in range(n_steps))
        bound = 2.0 * env[k] + REL * upd + 4 * EPS32 * np.abs(ref).max()
        bad = np.abs(w[k] - ref) > bound
        assert not bad.any(), (f"{k}: {int(bad.sum())} of {bad.size} elements off the reference trajectory, "
                               f"worst {np.abs(w[k] - ref)[bad].max():.3e} vs bound {bound[bad].min():.3e}")


@pytest.mark.parametrize("case", ["mf_bpr_sgd", "mf_bce_sgd", "mf_bpr_sgd_d300", "mf_bpr_sgd_reg"])
def test_sgd_touched_rows_mode(hip_device, case):
    """SGD restricted to the rows a batch touched: same result as the reference's dense step and
    every untouched row bit-identical (torch_engine.py:26-29, momentum 0)."""
    g = load_golden(case)
    n_steps = int(g["meta"][4])
    eng = engine_for_case(g, sgd_mode="rows")
    for s in range(n_steps):
        w_prev = params(g, f"w{s}")
        load_weights(eng, w_prev)
        loss, reg = eng.train_single_batch(batch_of(g, s))
        assert eng._rows_sgd
        assert_scalar_close(loss, g["losses"][s], what=f"loss step {s}")
        w = get_weights(eng)
        for k in KEYS:
            assert_step_close(w_prev[k], w[k], g[f"w{s + 1}/{k}"], 0.0, what=f"{k} step {s}")
        users = np.unique(g["users"][s])
        items = np.unique(g["items_a"][s]) if str(g["loss_kind"]) == "bce" else np.unique(
            np.concatenate([g["items_a"][s], g["third"][s].astype(np.int64)]))
        u_rest = np.setdiff1d(np.arange(w_prev["user_emb.weight"].shape[0]), users)
        i_rest = np.setdiff1d(np.arange(w_prev["item_emb.weight"].shape[0]), items)
        assert np.array_equal(w["user_emb.weight"][u_rest], w_prev["user_emb.weight"][u_rest])
        assert np.array_equal(w["item_emb.weight"][i_rest], w_prev["item_emb.weight"][i_rest])
        assert np.array_equal(w["user_bias.weight"][u_rest], w_prev["user_bias.weight"][u_rest])
        assert float(eng._g_flat.abs().max()) == 0.0


def test_predict_matches_reference(hip_device):
    g = load_golden("mf_predict")
    eng = make_engine(53, 47, 64, "sgd", "bpr", 0.1, 3)
    load_weights(eng, {k: g[f"w/{k}"] for k in KEYS})
    scores = eng.model.predict(g["users"], g["items"])  # numpy ids, as EvalEngine.predict passes
    assert isinstance(scores, torch.Tensor) and scores.device.type == "cuda"
    got = scores.flatten().to(torch.device("cpu")).detach().numpy()  # eval_engine.py:258-264
    assert_tensor_close(got, g["scores"], what="scores")
    fscores, reg = eng.model.forward((torch.from_numpy(g["users"]), torch.from_numpy(g["items"])))
    assert_tensor_close(fscores.cpu().numpy(), g["scores"], what="forward scores")
    _, ref_reg, _ = onp.mf_forward({k: g[f"w/{k}"] for k in KEYS}, g["users"], g["items"])
    assert_scalar_close(float(reg), float(ref_reg), what="forward regularizer")


@pytest.mark.parametrize("case", ["mf_epoch_adam", "mf_epoch_sgd"])
def test_epoch_through_dataloader_matches_reference(hip_device, case):
    """train_an_epoch fed with the reference's own kind of loader
    (DataLoader(PairwiseNegativeDataset, shuffle=True), base_data.py:247-253): same batch
    composition for the same torch seed, same add_scalar values, same final weights."""
    from torch.utils.data import DataLoader, Dataset

    class PairwiseNegativeDataset(Dataset):  # same fields as data/data_loaders.py:30-53
        def __init__(self, u, p, n):
            self.user_tensor, self.pos_item_tensor, self.neg_item_tensor = u, p, n

        def __getitem__(self, i):
            return self.user_tensor[i], self.pos_item_tensor[i], self.neg_item_tensor[i]

        def __len__(self):
            return self.user_tensor.size(0)

    g = load_golden(case)
    U, I, D, B, N, seed = (int(x) for x in g["meta"])
    opt = str(g["optimizer"])
    ds = PairwiseNegativeDataset(*(torch.LongTensor(g[k]) for k in ("users", "pos", "neg")))
    torch.manual_seed(seed)
    batches = [tuple(t.numpy() for t in b) for b in DataLoader(ds, batch_size=B, shuffle=True)]
    for mode in ("resident", "iterable"):
        eng = make_engine(U, I, D, opt, "bpr", 0.05, B)
        load_weights(eng, params(g, "w0"))
        loader = DataLoader(ds, batch_size=B, shuffle=True)
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()) as out:
            if mode == "resident":
                eng.train_an_epoch(loader, 0)
            else:  # any iterable of batches goes through the per-batch path
                eng.train_an_epoch(list(loader), 0)
        text = out.getvalue()
        assert "[Training Epoch 0], Loss" in text and "Execute [train_an_epoch]" in text
        scal = dict((t, v) for t, v, _ in eng.writer.scalars) if hasattr(eng.writer, "scalars") else None
        if scal is not None:
            assert_scalar_close(scal["model/loss"], float(g["scalar_loss"][0]), REL, "sum loss")
            assert_scalar_close(scal["model/regularizer"], float(g["scalar_reg"][0]), REL, "sum reg")
        # every element on the reference's trajectory (SGD: 1e-5 of the update; Adam: the derived envelope)
        assert_mf_end_state(get_weights(eng), params(g, "w0"), batches, opt, 0.05, f"{mode} epoch", ref=params(g, "w1"))


def test_epoch_per_batch_sequence(hip_device):
    """Feed the reference's exact batches one by one: per-batch (loss, reg) sequence."""
    g = load_golden("mf_epoch_sgd")
    U, I, D, B, N, _ = (int(x) for x in g["meta"])
    eng = make_engine(U, I, D, "sgd", "bpr", 0.05, B)
    load_weights(eng, params(g, "w0"))
    off = 0
    for j, bs in enumerate(g["batch_sizes"]):
        bt = g["batches"][:, off:off + bs]
        off += bs
        loss, reg = eng.train_single_batch(tuple(torch.from_numpy(bt[r]) for r in range(3)))
        assert_scalar_close(loss, g["losses"][j], REL, f"loss batch {j}")
        assert_scalar_close(reg, g["regs"][j], REL, f"reg batch {j}")
    w = get_weights(eng)
    for k in KEYS:
        assert_tensor_close(w[k], g[f"w1/{k}"], 1e-5, f"final {k}")


# ---- error behaviour ----------------------------------------------------------------------------

def test_out_of_range_index_raises_indexerror(hip_device):
    eng = make_engine(10, 10, 8, "sgd", "bpr", 0.1, 4)
    ok = (torch.tensor([1, 2]), torch.tensor([3, 4]), torch.tensor([5, 6]))
    eng.train_single_batch(ok)
    with pytest.raises(IndexError):
        eng.train_single_batch((torch.tensor([1, 10]), torch.tensor([3, 4]), torch.tensor([5, 6])))
    with pytest.raises(IndexError):
        eng.train_single_batch((torch.tensor([1, 2]), torch.tensor([3, -1]), torch.tensor([5, 6])))
    loss, _ = eng.train_single_batch(ok)  # the engine stays usable
    assert np.isfinite(loss)
    with pytest.raises(IndexError):
        eng.model.predict(np.array([0, 11]), np.array([0, 1]))


def test_batch_of_one_and_bad_loss(hip_device):
    eng = make_engine(10, 10, 8, "sgd", "bpr", 0.1, 4)
    with pytest.raises(IndexError):  # quirk Q4
        eng.train_single_batch((torch.tensor([1]), torch.tensor([3]), torch.tensor([5])))
    eng.loss = "hinge"
    with pytest.raises(RuntimeError, match="Unsupported loss type"):
        eng.train_single_batch((torch.tensor([1, 2]), torch.tensor([3, 4]), torch.tensor([5, 6])))


def test_checkpoint_roundtrip_on_device(hip_device, tmp_path):
    eng = make_engine(20, 15, 16, "adam", "bpr", 0.05, 4)
    batch = (torch.tensor([1, 2, 3]), torch.tensor([3, 4, 5]), torch.tensor([5, 6, 7]))
    eng.train_single_batch(batch)
    path = str(tmp_path / "mf.model")
    eng.save_checkpoint(path)
    sd = torch.load(path, map_location="cpu")
    assert list(sd.keys()) == list(KEYS)
    eng2 = make_engine(20, 15, 16, "adam", "bpr", 0.05, 4)
    with contextlib.redirect_stdout(io.StringIO()):
        eng2.resume_checkpoint(path)
    for k in KEYS:
        assert torch.equal(eng2.model.state_dict()[k].cpu(), sd[k])
    assert eng2.model.flat.is_cuda and eng2.model.user_emb.weight.data_ptr() == eng2.model.flat.data_ptr()


# ---- full size (BASELINE configs[1]: 6040 x 3706, dim 64, batch 4096) ----------------------------

C2 = dict(U=6040, I=3706, D=64, B=4096)


def c2_batch(seed):
    rng = np.random.default_rng(seed)
    U, I, B = C2["U"], C2["I"], C2["B"]
    p = 1.0 / np.arange(1, I + 1)
    p /= p.sum()
    users = rng.integers(0, U, B)
    pos = rng.permutation(I)[rng.choice(I, B, p=p)]
    neg = rng.integers(0, I, B)
    return users, pos, neg


def test_full_size_step_vs_oracle(hip_device):
    """One C2 batch: HIP loss / reg / dense gradients vs the numpy oracle, then an SGD step."""
    w0 = onp.init_params(C2["U"], C2["I"], C2["D"], seed=3)
    users, pos, neg = c2_batch(4)
    loss_ref, reg_ref, g_ref = onp.mf_bpr_grads(w0, users, pos, neg)
    eng = make_engine(C2["U"], C2["I"], C2["D"], "sgd", "bpr", 0.05, C2["B"])
    load_weights(eng, w0)
    batch = tuple(torch.from_numpy(a) for a in (users, pos, neg))
    loss, reg, grads = eng.backward_only(batch)
    assert_scalar_close(loss, loss_ref, what="loss")
    assert_scalar_close(reg, reg_ref, what="reg")
    for k in KEYS:
        assert_tensor_close(grads[k].cpu().numpy(), g_ref[k], what=f"grad {k}",
                            scale_floor=grad_scale_floor(k, C2["B"]))
    # conservation: every per-triple term d+ + d- lands once in user_bias, item_bias, global_bias
    gu = float(grads["user_bias.weight"].double().sum())
    gi = float(grads["item_bias.weight"].double().sum())
    gg = float(grads["global_bias"].double().sum())
    assert abs(gu - gg) < 1e-6 and abs(gi - gg) < 1e-6
    eng.train_single_batch(batch)
    w = get_weights(eng)
    w_ref = onp.copy_params(w0)
    onp.opt_step(w_ref, g_ref, onp.new_opt_state(w_ref, "sgd"), "sgd", 0.05)
    for k in KEYS:
        assert_step_close(w0[k], w[k], w_ref[k], 0.0, what=f"sgd {k}")
    untouched = np.setdiff1d(np.arange(C2["U"]), users)
    assert np.array_equal(w["user_emb.weight"][untouched], w0["user_emb.weight"][untouched])


@pytest.mark.parametrize("opt,lr", [("adam", 0.05), ("rmsprop", 0.01)])
def test_full_size_dense_optimizer_vs_oracle(hip_device, opt, lr):
    """Two chained C2 steps of a dense optimizer against the oracle (second step exercises
    non-zero moments on rows the batch does not touch)."""
    w_ref = onp.init_params(C2["U"], C2["I"], C2["D"], seed=5)
    eng = make_engine(C2["U"], C2["I"], C2["D"], opt, "bpr", lr, C2["B"])
    load_weights(eng, w_ref)
    st = onp.new_opt_state(w_ref, opt)
    for s in range(2):
        users, pos, neg = c2_batch(10 + s)
        w_prev = onp.copy_params(w_ref)
        st_prev = {k: ({kk: vv.copy() for kk, vv in v.items()} if isinstance(v, dict) else v)
                   for k, v in st.items()}
        loss_ref, reg_ref, g_ref = onp.mf_bpr_grads(w_ref, users, pos, neg)
        onp.opt_step(w_ref, g_ref, st, opt, lr)
        load_weights(eng, w_prev)  # isolate the step: start from the oracle's state
        eng.load_optimizer_state(s, st_prev.get("exp_avg"),
                                 st_prev.get("exp_avg_sq", st_prev.get("square_avg")))
        loss, reg = eng.train_single_batch(tuple(torch.from_numpy(a) for a in (users, pos, neg)))
        assert_scalar_close(loss, loss_ref, what=f"loss {s}")
        assert_scalar_close(reg, reg_ref, what=f"reg {s}")
        band = optimizer_band(w_prev, st_prev, g_ref, opt, lr, C2["B"])
        w = get_weights(eng)
        for k in KEYS:
            assert_step_close(w_prev[k], w[k], w_ref[k], band[k], what=f"{opt} {k} step {s}")


def test_full_size_epoch_resident_equals_per_batch(hip_device):
    """Idempotence of the batching layer at full size: the fused epoch driver (device-side
    permutation) and the per-batch path fed with the same permutation give the same weights."""
    import beta_recsys_amd as hp

    rng = np.random.default_rng(0)
    N = 3 * C2["B"] + 77
    users = rng.integers(0, C2["U"], N)
    pos = rng.integers(0, C2["I"], N)
    neg = rng.integers(0, C2["I"], N)
    w0 = onp.init_params(C2["U"], C2["I"], C2["D"], seed=9)
    gen = torch.Generator().manual_seed(123)
    batcher = hp.DeviceTripleBatcher(torch.from_numpy(users).cuda(), torch.from_numpy(pos).cuda(),
                                     torch.from_numpy(neg).cuda(), C2["B"], generator=gen)
    results = []
    for mode in ("resident", "list"):
        eng = make_engine(C2["U"], C2["I"], C2["D"], "sgd", "bpr", 0.05, C2["B"])
        load_weights(eng, w0)
        gen.manual_seed(123)
        with contextlib.redirect_stdout(io.StringIO()):
            eng.train_an_epoch(batcher if mode == "resident" else list(batcher), 0)
        results.append((get_weights(eng), dict((t, v) for t, v, _ in eng.writer.scalars)))
    (wa, sa), (wb, sb) = results
    for k in KEYS:
        assert_tensor_close(wa[k], wb[k], 1e-6, f"{k}")
    assert_scalar_close(sa["model/loss"], sb["model/loss"], 1e-6, "epoch loss")
    assert len(batcher) == 4


@pytest.mark.parametrize("optimizer", ["sgd", "adam", "rmsprop"])
def test_fused_epoch_matches_two_kernel_epoch(hip_device, optimizer):
    """The one-kernel-per-step epoch (update of step k-1 applied on the fly inside the gradient
    kernel of step k) evaluates the same expressions as grad kernel + dense optimizer sweep: the
    weights and moments agree to fp32 summation order of the atomics, for SGD the untouched rows bit
    for bit, and two epochs in a row leave every gradient buffer clean."""
    import beta_recsys_amd as hp

    rng = np.random.default_rng(2)
    N = 5 * C2["B"] + 1234  # six steps, the last one short
    users = torch.from_numpy(rng.integers(0, C2["U"], N)).cuda()
    pos = torch.from_numpy(rng.integers(0, 400, N)).cuda()   # few hot items: long merge runs
    neg = torch.from_numpy(rng.integers(0, C2["I"], N)).cuda()
    w0 = onp.init_params(C2["U"], C2["I"], C2["D"], seed=11)
    out = {}
    for fused in (True, False):
        eng = make_engine(C2["U"], C2["I"], C2["D"], optimizer, "bpr", 0.05, C2["B"])
        eng.fused_step = fused
        load_weights(eng, w0)
        batcher = hp.DeviceTripleBatcher(users, pos, neg, C2["B"], generator=torch.Generator().manual_seed(5))
        for epoch in range(2):
            batcher.generator.manual_seed(5 + epoch)
            with contextlib.redirect_stdout(io.StringIO()):
                eng.train_an_epoch(batcher, epoch)
        assert float(eng._g_flat.abs().max()) == 0.0
        if fused:
            for t in eng._fused_bufs["g"]:
                assert float(t.abs().max()) == 0.0
        st = eng.epoch_stats()
        assert st.step == 12
        out[fused] = (get_weights(eng), dict((t, v) for t, v, e in eng.writer.scalars if e == 1), st.loss,
                      {n: (None if getattr(eng.optimizer, n) is None else getattr(eng.optimizer, n).cpu().numpy().copy())
                       for n in ("exp_avg", "exp_avg_sq")})
    (wa, sa, la, oa), (wb, sb, lb, ob) = out[True], out[False]
    # The moments are linear / quadratic in the gradients: the two paths agree to 1e-5 of their scale.  The weights of
    # both paths are held to the oracle stepping the batches the batcher visited (its generator is seeded per epoch):
    # SGD every element within 1e-5 of the update, Adam / RMSprop every element inside the legal-trajectory envelope.
    for name in ("exp_avg", "exp_avg_sq"):
        if oa[name] is not None:
            assert_tensor_close(oa[name], ob[name], 1e-5, f"fused vs two-kernel {name}")
    data = [t.cpu().numpy() for t in (users, pos, neg)]
    batches = []
    for epoch in range(2):
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(5 + epoch)).numpy()
        batches += [tuple(d[perm[k:k + C2["B"]]] for d in data) for k in range(0, N, C2["B"])]
    traj = assert_mf_end_state(wa, w0, batches, optimizer, 0.05, "fused epochs")
    check_mf_end_state(wb, w0, batches, optimizer, 0.05, traj, "two-kernel epochs")
    assert_scalar_close(sa["model/loss"], sb["model/loss"], 1e-6, "epoch loss sum")
    assert_scalar_close(sa["model/regularizer"], sb["model/regularizer"], 1e-6, "epoch reg sum")
    assert_scalar_close(la, lb, 1e-5, "last loss")
    if optimizer == "sgd":
        never = np.setdiff1d(np.arange(C2["U"]), users.cpu().numpy())
        assert np.array_equal(wa["user_emb.weight"][never], w0["user_emb.weight"][never])


def test_bce_epoch_through_generic_loader(hip_device):
    """MF with loss 'bce' fed by a RatingDataset-style loader of (user, item, rating) batches
    (data/base_data.py:182-216): per-batch path, chained SGD steps vs the oracle."""
    U, I, D, B = 200, 150, 32, 64
    rng = np.random.default_rng(8)
    w = onp.init_params(U, I, D, seed=8)
    eng = make_engine(U, I, D, "sgd", "bce", 0.05, B)
    load_weights(eng, w)
    st = onp.new_opt_state(w, "sgd")
    batches, total = [], 0.0
    for k in range(5):
        n = B if k < 4 else 17
        b = (rng.integers(0, U, n), rng.integers(0, I, n), (rng.random(n) < 0.3).astype(np.float32))
        batches.append(tuple(torch.from_numpy(a) for a in b))
        loss, _ = onp.mf_train_step(w, st, b, "bce", "sgd", 0.05)
        total += loss
    with contextlib.redirect_stdout(io.StringIO()):
        eng.train_an_epoch(batches, 3)
    scal = dict((t, v) for t, v, e in eng.writer.scalars)
    assert_scalar_close(scal["model/loss"], total, REL, "epoch BCE loss")
    got = get_weights(eng)
    for k in KEYS:
        assert_tensor_close(got[k], w[k], 2e-6, f"bce epoch {k}")


def test_c4_shard_size_properties(hip_device):
    """One rank's share of BASELINE configs[3] (1.25M x 125k rows, dim 128, batch 65536): too big for
    the oracle in seconds, so size-independent properties: gradient conservation, every gradient row
    outside the batch exactly zero, touched-rows SGD leaves untouched rows bit-identical and the
    gradient buffer clean, and the step is linear in lr (two half steps of the gradient == one)."""
    U, I, D, B = 1_250_000, 125_000, 128, 65536
    eng = make_engine(U, I, D, "sgd", "bpr", 0.05, B)
    assert eng._setup() and eng._rows_sgd, "tables of this size take the touched-rows SGD path"
    gen = torch.Generator().manual_seed(1)
    users = torch.randint(0, U, (B,), generator=gen)
    pos = torch.randint(0, 2000, (B,), generator=gen)  # popular items: heavy merging
    neg = torch.randint(0, I, (B,), generator=gen)
    w0 = eng.model.flat.clone()
    loss, reg, grads = eng.backward_only((users, pos, neg))
    assert 0.3 < loss < 1.4 and reg > 0
    gg = float(grads["global_bias"].double().sum())
    assert abs(float(grads["user_bias.weight"].double().sum()) - gg) < 1e-5
    assert abs(float(grads["item_bias.weight"].double().sum()) - gg) < 1e-5
    touched_u = torch.zeros(U, dtype=torch.bool)
    touched_u[users] = True
    assert float(grads["user_emb.weight"][~touched_u.cuda()].abs().max()) == 0.0
    assert int((grads["user_emb.weight"].abs().sum(1) > 0).sum()) <= int(touched_u.sum())
    # one SGD step == w0 - lr * grad on the touched rows, nothing else moves
    eng.train_single_batch((users, pos, neg))
    w1 = eng.model.flat
    g_flat = torch.cat([grads[k].reshape(-1) for k in
                        ("user_emb.weight", "item_emb.weight", "user_bias.weight", "item_bias.weight", "global_bias")])
    expect = w0 - 0.05 * g_flat
    err = float((w1 - expect).abs().max())
    assert err <= 1e-5 * float((0.05 * g_flat).abs().max()) + 1e-7, err
    assert torch.equal(w1[: U * D].view(U, D)[~touched_u.cuda()], w0[: U * D].view(U, D)[~touched_u.cuda()])
    assert float(eng._g_flat.abs().max()) == 0.0


def test_c1_ml100k_shaped_epoch_vs_cpu_port(hip_device):
    """BASELINE configs[0] (mf_default.json shape): 943 users x 1682 items, dim 64, batch 400, adam
    lr 0.05, one epoch of ~98k triples through a real DataLoader(shuffle=True) — the HIP engine vs the
    reference's PyTorch-CPU op sequence (oracle/torch_port.py) fed with the very same batches."""
    from torch.utils.data import DataLoader, Dataset

    from oracle.torch_port import TorchMFPort

    class PairwiseNegativeDataset(Dataset):
        def __init__(self, u, p, n):
            self.user_tensor, self.pos_item_tensor, self.neg_item_tensor = u, p, n

        def __getitem__(self, i):
            return self.user_tensor[i], self.pos_item_tensor[i], self.neg_item_tensor[i]

        def __len__(self):
            return self.user_tensor.size(0)

    U, I, D, B, N = 943, 1682, 64, 400, 98114
    rng = np.random.default_rng(2020)
    p = 1.0 / np.arange(1, I + 1)
    users = torch.from_numpy(rng.integers(0, U, N))
    pos = torch.from_numpy(rng.permutation(I)[rng.choice(I, N, p=p / p.sum())])
    neg = torch.from_numpy(rng.integers(0, I, N))
    loader = DataLoader(PairwiseNegativeDataset(users, pos, neg), batch_size=B, shuffle=True)
    w0 = onp.init_params(U, I, D, seed=1)
    eng = make_engine(U, I, D, "adam", "bpr", 0.05, B)
    load_weights(eng, w0)
    torch.manual_seed(99)
    with contextlib.redirect_stdout(io.StringIO()):
        eng.train_an_epoch(loader, 0)
    scal = dict((t, v) for t, v, _ in eng.writer.scalars)
    # the same batches for the CPU port: same seed -> same DataLoader order
    port = TorchMFPort(w0, "adam", 0.05, "bpr")
    torch.manual_seed(99)
    tot_loss = tot_reg = 0.0
    n_batches = 0
    for batch in loader:
        l, r = port.step(batch)
        tot_loss += l
        tot_reg += r
        n_batches += 1
    assert n_batches == 246 and N % B != 1
    # 246 chained Adam steps: the per-step losses of two correct implementations drift apart as their weights do (an
    # element whose gradient is ~eps moves by up to lr either way, helpers.optimizer_band).  The bounds are therefore
    # taken from the oracle itself: the spread of the epoch sums and of the end weights over runs whose gradients are
    # moved by what north_star allows (1e-5 of their scale) -- twice that spread, every element, no outliers; both
    # the engine and the CPU port must lie inside.
    torch.manual_seed(99)
    batches = [tuple(t.numpy() for t in batch) for batch in loader]
    w_ref, env, upd, (sums_ref, sums_spread) = mf_trajectory(w0, batches, "adam", 0.05, trials=4, with_sums=True)
    for who, weights, sums in (("engine", get_weights(eng), (scal["model/loss"], scal["model/regularizer"])),
                               ("CPU port", port.numpy_weights(), (tot_loss, tot_reg))):
        assert_on_trajectory(weights, w_ref, env, upd, f"C1 epoch through the DataLoader, {who}", pool=True)
        for j, name in enumerate(("loss", "regularizer")):
            tol = 1e-5 * abs(sums_ref[j]) + 2 * sums_spread[j]
            assert abs(sums[j] - sums_ref[j]) <= tol, (
                f"{who}: epoch {name} sum {sums[j]!r} vs the oracle's {sums_ref[j]!r}: off by "
                f"{abs(sums[j] - sums_ref[j]):.3e} > 1e-5 relative + twice the legal spread {sums_spread[j]:.3e}")


def test_bce_resident_epoch_through_rating_dataloader(hip_device):
    """loss 'bce' fed by DataLoader(RatingDataset) (data/base_data.py:182-216): the resident epoch
    driver (hiprec_mf_bce_epoch) draws the loader's own order and matches the oracle fed with the
    same batches."""
    from torch.utils.data import DataLoader, Dataset

    class RatingDataset(Dataset):  # same fields as data/data_loaders.py:4-27
        def __init__(self, u, i, r):
            self.user_tensor, self.item_tensor, self.target_tensor = u, i, r

        def __getitem__(self, k):
            return self.user_tensor[k], self.item_tensor[k], self.target_tensor[k]

        def __len__(self):
            return self.user_tensor.size(0)

    U, I, D, B, N = 200, 150, 32, 300, 1000
    rng = np.random.default_rng(9)
    users = torch.from_numpy(rng.integers(0, U, N))
    items = torch.from_numpy(rng.integers(0, 40, N))
    ratings = torch.from_numpy((rng.random(N) < 0.3).astype(np.float32))
    loader = DataLoader(RatingDataset(users, items, ratings), batch_size=B, shuffle=True)
    for opt, lr in (("sgd", 0.05), ("adam", 0.01)):
        w = onp.init_params(U, I, D, seed=9)
        eng = make_engine(U, I, D, opt, "bce", lr, B)
        load_weights(eng, w)
        assert eng.prepare_epoch(loader) is not None
        torch.manual_seed(4)
        with contextlib.redirect_stdout(io.StringIO()):
            eng.train_an_epoch(loader, 0)
        st = onp.new_opt_state(w, opt)
        w_start = onp.copy_params(w)
        torch.manual_seed(4)
        total, batches = 0.0, []
        for bu, bi, br in loader:
            batches.append((bu.numpy(), bi.numpy(), br.numpy()))
            loss, _ = onp.mf_train_step(w, st, batches[-1], "bce", opt, lr)
            total += loss
        scal = dict((t, v) for t, v, _ in eng.writer.scalars)
        assert_scalar_close(scal["model/loss"], total, REL, f"{opt} epoch BCE loss")
        assert_mf_end_state(get_weights(eng), w_start, batches, opt, lr, f"{opt} BCE epoch", loss="bce")


@pytest.mark.parametrize("batch", [16, 400, 1000, 4096, 8192])
@pytest.mark.parametrize("third_kind", ["int64", "float32"])
def test_stage_epoch_kernel(hip_device, batch, third_kind):
    """The native batcher: batch composition is exactly perm[b*bs:(b+1)*bs] (as a multiset of whole
    triples), every batch comes out sorted by item, the last batch is short."""
    from beta_recsys_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(batch)
    n = 3 * batch + batch // 3 + 1
    users = rng.integers(0, 10_000, n)
    # float32 case: a small catalogue (32-bit (item, position) keys); int64 case: ids beyond 2^19 (64-bit keys)
    items = rng.integers(0, 500, n) if third_kind == "float32" else rng.integers(0, 2**31 - 1, n) // rng.choice([1, 4096], n)
    third = rng.integers(0, 10_000, n) if third_kind == "int64" else rng.random(n).astype(np.float32)
    perm = rng.permutation(n)
    tu, ti, tt, tp = (torch.from_numpy(a).cuda() for a in (users, items, third, perm))
    for use_perm in (True, False):
        ou, oi, ot = torch.zeros_like(tu), torch.zeros_like(ti), torch.zeros_like(tt)
        _lib.check(lib.hiprec_stage_epoch(
            _lib.ptr(tu), _lib.ptr(ti), _lib.ptr(tt), tt.element_size(), _lib.ptr(tp) if use_perm else None,
            n, batch, _lib.ptr(ou), _lib.ptr(oi), _lib.ptr(ot), _lib.stream_ptr(hip_device)))
        gu, gi, gt = ou.cpu().numpy(), oi.cpu().numpy(), ot.cpu().numpy()
        order = perm if use_perm else np.arange(n)
        for b0 in range(0, n, batch):
            sl = slice(b0, min(n, b0 + batch))
            src = order[sl]
            assert np.all(np.diff(gi[sl]) >= 0), "batch must be sorted by item"
            want = sorted(zip(items[src].tolist(), users[src].tolist(), third[src].tolist()))
            got = sorted(zip(gi[sl].tolist(), gu[sl].tolist(), gt[sl].tolist()))
            assert want == got, "batch must contain exactly its triples"


@pytest.mark.parametrize("n", [1, 2, 3, 1000, 4097, 1_000_003])
def test_random_permutation_kernel(hip_device, n):
    """The device shuffle is a bijection of [0, n), depends on the seed, and mixes well."""
    from beta_recsys_amd import _lib

    lib = _lib.load()
    outs = []
    for seed in (1, 2):
        out = torch.full((n,), -1, dtype=torch.int64, device=hip_device)
        _lib.check(lib.hiprec_random_permutation(_lib.ptr(out), n, seed, _lib.stream_ptr(hip_device)))
        p = out.cpu().numpy()
        assert np.array_equal(np.sort(p), np.arange(n)), "must be a permutation"
        outs.append(p)
    if n >= 1000:
        assert (outs[0] != outs[1]).mean() > 0.99
        assert (outs[0] == np.arange(n)).mean() < 0.01
        # neighbours are sent far apart and without a preferred direction
        d = np.diff(outs[0].astype(np.float64))
        assert abs(np.corrcoef(outs[0][:-1], outs[0][1:])[0, 1]) < 0.05
        assert abs((d > 0).mean() - 0.5) < 0.05


@pytest.mark.parametrize("D,optimizer,reg", [(100, "adam", None), (200, "rmsprop", None), (256, "adam", 0.01),
                                              (4, "sgd", None), (300, "adam", None)])
def test_fused_epoch_other_widths_against_the_oracle(hip_device, D, optimizer, reg):
    """The fused step kernel's other instantiations (2 and 4 columns per lane, non-multiples of 64, the
    regularised loss) and the width (300 > 256) that must fall back to the two-kernel epoch: five
    sequential batches vs the numpy oracle stepping the same batches."""
    import beta_recsys_amd as hp

    U, I, B = 500, 300, 96
    rng = np.random.default_rng(D)
    n = 4 * B + 17
    triples = (rng.integers(0, U, n), rng.integers(0, 40, n), rng.integers(0, I, n))
    w = onp.init_params(U, I, D, seed=D)
    w_start = onp.copy_params(w)
    eng = make_engine(U, I, D, optimizer, "bpr", 0.02, B, reg=reg)
    load_weights(eng, w)
    loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a).cuda() for a in triples), B, shuffle=False)
    eng._setup()
    assert eng._fused_ok(None) == (D <= 256)
    st = onp.new_opt_state(w, optimizer)
    total = 0.0
    for k in range(0, n, B):
        loss, _ = onp.mf_train_step(w, st, tuple(a[k:k + B] for a in triples), "bpr", optimizer, 0.02,
                                    reg_coef=0.0 if reg is None else reg)
        total += loss
    with contextlib.redirect_stdout(io.StringIO()):
        eng.train_an_epoch(loader, 0)
    stats = eng.epoch_stats()
    assert stats.step == 5
    assert_scalar_close(stats.loss_sum, total, 1e-5, "epoch loss sum")
    got = get_weights(eng)
    batches = [tuple(a[k:k + B] for a in triples) for k in range(0, n, B)]
    assert_mf_end_state(got, w_start, batches, optimizer, 0.02, f"fused epoch D={D}",
                        reg_coef=0.0 if reg is None else reg)


def test_fused_epoch_flags_out_of_range_ids_and_handles_empty_epochs(hip_device):
    import beta_recsys_amd as hp

    U, I, D, B = 50, 40, 64, 16
    eng = make_engine(U, I, D, "adam", "bpr", 0.01, B)
    w0 = get_weights(eng)
    empty = hp.DeviceTripleBatcher(*(torch.zeros(0, dtype=torch.int64).cuda() for _ in range(3)), B, shuffle=False)
    with contextlib.redirect_stdout(io.StringIO()):
        eng.train_an_epoch(empty, 0)
    assert eng.epoch_stats().step == 0
    for k in KEYS:
        assert np.array_equal(get_weights(eng)[k], w0[k])
    users = torch.arange(2 * B) % U
    users[5] = U + 3                                       # nn.Embedding would raise IndexError
    bad = hp.DeviceTripleBatcher(users.cuda(), (torch.arange(2 * B) % I).cuda(), (torch.arange(2 * B) % I).cuda(),
                                 B, shuffle=False)
    with pytest.raises(IndexError), contextlib.redirect_stdout(io.StringIO()):
        eng.train_an_epoch(bad, 1)


def test_fused_epoch_with_batches_larger_than_the_gather_grid(hip_device):
    """Batches beyond 256 blocks x 16 waves make every gather block loop (three trips here, the last one
    ragged) and publish one partial per block for all of them: loss and weights vs the oracle."""
    import beta_recsys_amd as hp

    U, I, D, B = 3000, 2000, 64, 10000
    rng = np.random.default_rng(17)
    n = 2 * B + 777
    triples = (rng.integers(0, U, n), rng.integers(0, 200, n), rng.integers(0, I, n))
    w = onp.init_params(U, I, D, seed=17)
    eng = make_engine(U, I, D, "sgd", "bpr", 0.05, B)
    load_weights(eng, w)
    loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a).cuda() for a in triples), B, shuffle=False)
    st = onp.new_opt_state(w, "sgd")
    total = sum(onp.mf_train_step(w, st, tuple(a[k:k + B] for a in triples), "bpr", "sgd", 0.05)[0]
                for k in range(0, n, B))
    with contextlib.redirect_stdout(io.StringIO()):
        eng.train_an_epoch(loader, 0)
    stats = eng.epoch_stats()
    assert stats.step == 3
    assert_scalar_close(stats.loss_sum, total, 1e-5, "epoch loss sum")
    got = get_weights(eng)
    for k in KEYS:
        assert_tensor_close(got[k], w[k], 1e-5, f"large-batch fused epoch {k}", scale_floor=grad_scale_floor(k, B) * 0.05)


def test_c1_config_against_the_reference_run(hip_device):
    """BASELINE configs[0] (mf_default.json shape, adam lr 0.05, batch 400): the engine, built with the
    reference's seed, reproduces the losses of the real reference's own run step by step and ends with
    the same weight checksums."""
    g = load_golden("mf_c1_adam")
    U, I, D, B, n_steps, seed = (int(x) for x in g["meta"])
    torch.manual_seed(seed)
    eng = make_engine(U, I, D, "adam", "bpr", 0.05, B)
    w_init = get_weights(eng)
    for s in range(n_steps):
        loss, reg = eng.train_single_batch(tuple(torch.from_numpy(g[k][s]) for k in ("users", "pos", "neg")))
        assert_scalar_close(loss, g["losses"][s], 1e-5, f"loss of step {s}")
        assert_scalar_close(reg, g["regs"][s], 1e-5, f"regularizer of step {s}")
    got = get_weights(eng)
    # the golden holds the reference's end state as per-tensor sums of squares and the first 64 weights; the oracle
    # (pinned on the other goldens) supplies the full trajectory from the same initial weights: every element of the
    # engine inside the legal envelope, and the reference's own sampled weights and checksums inside it too
    batches = [tuple(g[k][s] for k in ("users", "pos", "neg")) for s in range(n_steps)]
    w_ref, env, upd = assert_mf_end_state(got, w_init, batches, "adam", 0.05, "C1 run")
    for k in KEYS:
        ref = w_ref[k].astype(np.float64).reshape(-1)
        bound = (2.0 * env[k] + REL * upd[k] + 4 * EPS32 * np.abs(ref).max()).reshape(-1)
        head = g[f"head/{k}"].astype(np.float64)
        bad = np.abs(head - ref[:head.size]) > bound[:head.size]
        assert not bad.any(), f"{k}: {int(bad.sum())} of the reference's sampled end weights are off the oracle's trajectory"
        bad = np.abs(got[k].reshape(-1)[:head.size] - head) > 2 * bound[:head.size]
        assert not bad.any(), f"{k}: {int(bad.sum())} sampled weights differ from the reference's beyond both envelopes"
        tol = float((2 * np.abs(ref) * bound + bound * bound).sum())
        a = got[k].astype(np.float64)
        assert abs((a * a).sum() - float(g[f"sumsq/{k}"])) <= 2 * tol + 1e-12, (k, (a * a).sum(), float(g[f"sumsq/{k}"]), tol)


@pytest.mark.parametrize("batch", [64, 1000, 4096])
def test_stage_epoch_shuffled_equals_permutation_then_stage(hip_device, batch):
    """The shuffle folded into the staging kernel (P_seed evaluated on the fly) lays the epoch out exactly as
    hiprec_random_permutation(seed) followed by hiprec_stage_epoch(perm) does -- bit for bit."""
    from beta_recsys_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(batch)
    n = 5 * batch + 17
    tu, ti, tt = (torch.from_numpy(rng.integers(0, 3000, n)).cuda() for _ in range(3))
    st = _lib.stream_ptr(hip_device)
    for seed in (1, 0x1234_5678_9ABC_DEF0):
        perm = torch.empty(n, dtype=torch.int64, device=hip_device)
        _lib.check(lib.hiprec_random_permutation(_lib.ptr(perm), n, seed, st))
        a = [torch.zeros_like(tu) for _ in range(3)]
        b = [torch.zeros_like(tu) for _ in range(3)]
        _lib.check(lib.hiprec_stage_epoch(_lib.ptr(tu), _lib.ptr(ti), _lib.ptr(tt), 8, _lib.ptr(perm), n, batch,
                                          *(_lib.ptr(x) for x in a), st))
        _lib.check(lib.hiprec_stage_epoch_shuffled(_lib.ptr(tu), _lib.ptr(ti), _lib.ptr(tt), 8, seed, n, batch,
                                                   *(_lib.ptr(x) for x in b), st))
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        # and it IS a shuffle of whole triples: same multiset, not the identity order
        got = sorted(zip(*(x.cpu().tolist() for x in b)))
        want = sorted(zip(tu.cpu().tolist(), ti.cpu().tolist(), tt.cpu().tolist()))
        assert got == want


def test_next_epoch_prefetch_is_the_same_training_run(hip_device):
    """train_an_epoch stages the NEXT epoch on a side stream while the current one trains (TrainEngine._train,
    core/train_engine.py:225-240, passes the same loader every epoch).  With the same torch seed the shuffle
    keys are drawn in the same order, so three epochs with and without the prefetch visit the same batches:
    same per-epoch loss sums, same weights (up to the order of the fp32 atomics)."""
    import beta_recsys_amd as hp

    U, I, D, B, N = 500, 300, 64, 256, 256 * 9 + 40
    rng = np.random.default_rng(3)
    triples = [torch.from_numpy(rng.integers(0, hi, N)).cuda() for hi in (U, I, I)]
    w0 = onp.init_params(U, I, D, seed=3)
    runs = []
    for prefetch in (True, False):
        eng = make_engine(U, I, D, "adam", "bpr", 0.01, B, prefetch_epoch=prefetch)
        load_weights(eng, w0)
        loader = hp.DeviceTripleBatcher(*triples, B)
        torch.manual_seed(11)
        sums = []
        for epoch in range(3):
            with contextlib.redirect_stdout(io.StringIO()):
                eng.train_an_epoch(loader, epoch)
            sums.append(eng.epoch_stats().loss_sum)
            if prefetch:
                assert getattr(eng, "_prefetched", None) is not None and eng._prefetched[0] is loader
        # a different loader object must not be served the pending prefetch
        other = hp.DeviceTripleBatcher(*triples, B, shuffle=False)
        prepared = eng.prepare_epoch(other)
        assert torch.equal(prepared[1][:B].sort().values, triples[1][:B].sort().values)
        assert getattr(eng, "_prefetched", None) is None
        runs.append((sums, get_weights(eng)))
    (sa, wa), (sb, wb) = runs
    for a, b in zip(sa, sb):
        assert_scalar_close(a, b, 1e-5, "epoch loss sum with / without prefetch")
    # both runs visited the same three shuffles (the seeds come from torch's CPU generator): replay the draws and
    # hold both end states to the oracle's trajectory over those 30 batches, every element
    torch.manual_seed(11)
    data = [t.cpu().numpy() for t in triples]
    batches = []
    for _ in range(3):
        perm = loader.permutation().cpu().numpy()
        batches += [tuple(d[perm[k:k + B]] for d in data) for k in range(0, N, B)]
    traj = assert_mf_end_state(wa, w0, batches, "adam", 0.01, "3 epochs with prefetch", trials=16)
    check_mf_end_state(wb, w0, batches, "adam", 0.01, traj, "3 epochs without prefetch")


def test_prefetched_epoch_is_dropped_when_the_loader_data_changed(hip_device):
    """the prefetched staging is keyed on the loader's tensors (address, length, in-place version), not
    only on the loader object: a caller that resamples the negatives in place between two epochs trains the next
    epoch on the NEW negatives."""
    import beta_recsys_amd as hp

    U, I, D, B, N = 400, 50, 16, 128, 128 * 4
    rng = np.random.default_rng(5)
    users, pos = (torch.from_numpy(rng.integers(0, hi, N)).cuda() for hi in (U, I))
    neg = torch.zeros(N, dtype=torch.int64, device="cuda")          # epoch 0: every negative is item 0
    eng = make_engine(U, I, D, "sgd", "bpr", 0.05, B)
    loader = hp.DeviceTripleBatcher(users, pos, neg, B)
    with contextlib.redirect_stdout(io.StringIO()):
        eng.train_an_epoch(loader, 0)
    assert eng._prefetched is not None                               # epoch 1 is staged (with negatives = 0)
    neg.fill_(I - 1)                                                  # resampled IN PLACE
    w_before = get_weights(eng)["item_emb.weight"].copy()
    staged = eng.prepare_epoch(loader)
    assert bool((staged[2] == I - 1).all()), "the stale staging was served"
    eng.run_prepared_epoch(staged)
    w_after = get_weights(eng)["item_emb.weight"]
    assert not np.array_equal(w_after[I - 1], w_before[I - 1]), "the new negative item's row was never touched"


def test_engine_on_a_device_that_is_not_current(hip_device):
    """the reference never calls set_device and TrainEngine.get_device hands out 'cuda:N'; every
    libhiprec launch must go to the device its tensors live on even while another device is current."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    g = load_golden("mf_bpr_adam")
    assert torch.cuda.current_device() == 0
    eng = engine_for_case(g, device_str="cuda:1")
    w0 = params(g, "w0")
    load_weights(eng, w0)
    loss, reg = eng.train_single_batch(batch_of(g, 0))
    assert torch.cuda.current_device() == 0
    assert_scalar_close(loss, g["losses"][0], what="loss on cuda:1 while cuda:0 is current")


@pytest.mark.parametrize("optimizer,fused", [("adam", True), ("sgd", True), ("adam", False)])
def test_epoch_enqueued_in_pieces_equals_the_whole_epoch(hip_device, optimizer, fused):
    """run_prepared_epoch(steps=(a, b)): an epoch enqueued piece by piece (bench.py puts timestamps between the
    pieces) is the same training run as the epoch enqueued at once -- the fused driver carries the pending
    update and the buffer rotation from piece to piece, the last piece flushes."""
    import beta_recsys_amd as hp

    U, I, D, B, N = 400, 250, 64, 128, 128 * 11 + 50   # 12 steps, last batch short
    rng = np.random.default_rng(5)
    triples = [torch.from_numpy(rng.integers(0, hi, N)).cuda() for hi in (U, I, I)]
    w0 = onp.init_params(U, I, D, seed=5)
    out = []
    for pieces in (None, [(0, 1), (1, 5), (5, 5), (5, 11), (11, 12)]):
        eng = make_engine(U, I, D, optimizer, "bpr", 0.02, B)
        eng.fused_step = fused
        load_weights(eng, w0)
        prepared = eng.prepare_epoch(hp.DeviceTripleBatcher(*triples, B, shuffle=False))
        if pieces is None:
            eng.run_prepared_epoch(prepared, sync=False)
        else:
            for piece in pieces:
                eng.run_prepared_epoch(prepared, sync=False, steps=piece)
        st = eng.epoch_stats()
        assert st.step == 12
        out.append((st.loss, get_weights(eng)))
    with pytest.raises(ValueError):
        eng.run_prepared_epoch(prepared, sync=False, steps=(3, 13))
    (la, wa), (lb, wb) = out
    assert_scalar_close(la, lb, 1e-6, "last-step loss, whole epoch vs pieces")
    # the two runs differ in the order of their fp32 atomics only; both are held to the oracle's 12 steps
    data = [t.cpu().numpy() for t in triples]
    batches = [tuple(d[k:k + B] for d in data) for k in range(0, N, B)]
    traj = assert_mf_end_state(wa, w0, batches, optimizer, 0.02, "whole epoch")
    check_mf_end_state(wb, w0, batches, optimizer, 0.02, traj, "epoch in pieces")


# ---- owned-rows SGD step (csrc/mf_owned.hip): the HBM-resident regime of BASELINE configs[3] ----------

def _zipf_triples(rng, n, U, I, hot=None):
    users = rng.integers(0, U, n)
    p = 1.0 / np.arange(1, I + 1)
    pos = rng.permutation(I)[rng.choice(I, n, p=p / p.sum())]
    if hot is not None:
        pos[rng.random(n) < hot] = 7
    neg = rng.integers(0, I, n)
    return users, pos, neg


@pytest.mark.parametrize("U,I,D,B,steps,hot,reg", [
    (300, 200, 64, 256, 5, None, None),     # everything shared: tiny tables, every row occurs many times
    (50_000, 3_000, 128, 2048, 4, 0.5, None),  # mostly single users, one item holding half of the positives
    (5_000, 4_000, 100, 1000, 3, None, 0.01),  # dim not a multiple of 64, short last batch, reg != 0
    (2_000, 1_500, 256, 512, 3, 0.1, None),
])
@pytest.mark.parametrize("form", ["owned", "owned_atomic"])
def test_owned_rows_epoch_matches_the_oracle(hip_device, U, I, D, B, steps, hot, reg, form):
    """sgd_mode 'owned' (owner pulls: a gradient launch that stores the parts of shared rows plainly + one wave per
    shared row that sums them; no float atomics) and 'owned_atomic' (one launch per step, shared rows collect
    device-scope atomic adds): rows updated in place by whoever holds their complete gradient.
    Against oracle/mf_numpy.py step by step (mf.py:92-119 + torch.optim.SGD): per-epoch loss sum, every weight
    within 1e-5 of the update scale, rows the epoch never touched bit-identical, accumulators left clean; and
    the same epoch enqueued in pieces."""
    import beta_recsys_amd as hp

    rng = np.random.default_rng(U + B)
    n = steps * B - (B // 3 if D == 100 else 0)
    users, pos, neg = _zipf_triples(rng, n, U, I, hot)
    lr = 0.05
    w0 = onp.init_params(U, I, D, seed=3)
    triples = [torch.from_numpy(a).cuda() for a in (users, pos, neg)]
    # the oracle on the same batches (the batcher only reorders INSIDE a batch)
    w = onp.copy_params(w0)
    st = onp.new_opt_state(w, "sgd")
    total = 0.0
    for k in range(0, n, B):
        loss, _ = onp.mf_train_step(w, st, (users[k:k + B], pos[k:k + B], neg[k:k + B]), "bpr", "sgd", lr,
                                    reg_coef=reg or 0.0)
        total += loss
    for pieces in (None, [(0, 1), (1, 2), (2, steps)]):
        eng = make_engine(U, I, D, "sgd", "bpr", lr, B, reg=reg, sgd_mode=form)
        load_weights(eng, w0)
        prepared = eng.prepare_epoch(hp.DeviceTripleBatcher(*triples, B, shuffle=False))
        assert prepared.own is not None and eng._setup() and eng._owned_sgd
        assert isinstance(prepared.own, hp.mf.RowContributions) == (form == "owned")
        if pieces is None:
            eng.run_prepared_epoch(prepared, sync=False)
        else:
            for piece in pieces:
                eng.run_prepared_epoch(prepared, sync=False, steps=piece)
        stt = eng.epoch_stats()
        assert stt.step == steps
        assert_scalar_close(stt.loss_sum, total, REL, "epoch loss sum")
        got = get_weights(eng)
        for k in KEYS:
            assert_update_close(w0[k], got[k], w[k], what=f"{k} after {steps} owned-rows steps")
        for k, rows in (("user_emb.weight", users), ("item_emb.weight", np.concatenate([pos, neg]))):
            untouched = np.ones(w0[k].shape[0], dtype=bool)
            untouched[rows] = False
            assert np.array_equal(got[k][untouched], w0[k][untouched]), f"{k}: untouched rows moved"
        if form == "owned_atomic":
            assert float(eng._owned_bufs["acc"].abs().max()) == 0.0 and int(eng._owned_bufs["arrived"].abs().max()) == 0


def test_owned_rows_step_at_c4_shard_size(hip_device):
    """One rank's share of BASELINE configs[3] (1.25M x 125k rows, dim 128, batch 65536, Zipf positives): the
    owned-rows step and the two-kernel touched-rows path (dense gradient buffer) are the same SGD step -- equal
    weights on every touched row to 1e-5 of the update, every other row bit-identical -- over three steps, and
    the accumulators come back clean."""
    import beta_recsys_amd as hp

    U, I, D, B, steps = 1_250_000, 125_000, 128, 65536, 3
    rng = np.random.default_rng(11)
    users, pos, neg = _zipf_triples(rng, steps * B, U, I)
    triples = [torch.from_numpy(a).cuda() for a in (users, pos, neg)]
    out = {}
    for mode in ("rows", "owned", "owned_atomic"):
        torch.manual_seed(5)
        eng = make_engine(U, I, D, "sgd", "bpr", 0.05, B, sgd_mode=mode)
        w0 = eng.model.flat.clone()
        prepared = eng.prepare_epoch(hp.DeviceTripleBatcher(*triples, B, shuffle=False))
        assert (prepared.own is not None) == (mode != "rows")
        st = eng.run_prepared_epoch(prepared)
        assert st.step == steps and 0.3 < st.loss < 1.4
        out[mode] = (eng.model.flat.clone(), st.loss_sum)
        if mode == "owned_atomic":
            assert float(eng._owned_bufs["acc"].abs().max()) == 0.0
            assert int(eng._owned_bufs["arrived"].abs().max()) == 0
        del eng
    (wa, la) = out["rows"]
    upd = float((wa - w0).abs().max())
    for mode in ("owned", "owned_atomic"):
        wb, lb = out[mode]
        assert_scalar_close(lb, la, 1e-5, f"loss sum, {mode} vs touched-rows path")
        assert float((wa - wb).abs().max()) <= 1e-5 * upd + 4 * 1.2e-7 * float(w0.abs().max()), mode
    wb = out["owned"][0]
    moved = (wb != w0) | (out["owned_atomic"][0] != w0)
    touched = torch.zeros_like(moved)
    tu = torch.from_numpy(np.unique(users)).cuda()
    ti = torch.from_numpy(np.unique(np.concatenate([pos, neg]))).cuda()
    touched[: U * D].view(U, D)[tu] = True
    touched[U * D:(U + I) * D].view(I, D)[ti] = True
    touched[(U + I) * D:(U + I) * D + U][tu] = True
    touched[(U + I) * D + U:(U + I) * D + U + I][ti] = True
    touched[-1] = True
    assert not bool((moved & ~touched).any()), "a row outside the batches moved"
    # ... and each form against the ORACLE at this size, on the compacted problem: SGD neither reads nor
    # writes a row the three batches do not name, so oracle/mf_numpy.py runs on the touched rows renumbered
    from helpers import assert_sgd_exact

    uu, u_inv = np.unique(users, return_inverse=True)
    ui, i_inv = np.unique(np.concatenate([pos, neg]), return_inverse=True)

    def compact(flat):
        o = np.cumsum([0, U * D, I * D, U, I, 1])
        ue, ie = flat[o[0]:o[1]].view(U, D), flat[o[1]:o[2]].view(I, D)
        return {"user_emb.weight": ue[tu].cpu().numpy(), "item_emb.weight": ie[ti].cpu().numpy(),
                "user_bias.weight": flat[o[2]:o[3]][tu].cpu().numpy().reshape(-1, 1),
                "item_bias.weight": flat[o[3]:o[4]][ti].cpu().numpy().reshape(-1, 1),
                "global_bias": flat[o[4]:o[5]].cpu().numpy().copy()}

    wc0 = compact(w0)
    w = onp.copy_params(wc0)
    st = onp.new_opt_state(w, "sgd")
    ref_loss = 0.0
    for k in range(steps):
        sl = slice(k * B, (k + 1) * B)
        loss, _ = onp.mf_train_step(w, st, (u_inv[sl], i_inv[:steps * B][sl], i_inv[steps * B:][sl]), "bpr", "sgd", 0.05)
        ref_loss += loss
    for mode in ("owned", "owned_atomic", "rows"):
        assert_scalar_close(out[mode][1], ref_loss, 1e-5, f"loss sum, {mode} vs the oracle on the compacted problem")
        assert_sgd_exact(compact(out[mode][0]), w, wc0, f"configs[3] shard size, {mode}, touched rows", lr=0.05, batch=B)


@pytest.mark.parametrize("n,bs,U,I", [(1000, 128, 50, 30), (3 * 4096 + 77, 4096, 100_000, 2_000), (65536, 65536, 1_250_000, 125_000)])
def test_batch_row_ownership_kernel(hip_device, n, bs, U, I):
    """hiprec_batch_row_ownership (one hash table per batch, the table position is the slot) states the same
    contract as the sort-based batch_row_ownership_torch: same single / shared classification, same occurrence
    counts, one slot per shared row."""
    from beta_recsys_amd.mf import batch_row_ownership, batch_row_ownership_torch
    from test_host_logic import _brute_force_ownership_check

    rng = np.random.default_rng(n)
    users, pos, neg = _zipf_triples(rng, n, U, I)
    users[3], pos[min(77, n - 1)] = U, -1
    tu, tp, tn = (torch.from_numpy(a).cuda() for a in (users, pos, neg))
    own, total, stride = batch_row_ownership(tu, tp, tn, bs, U, I)
    own_t, total_t, _ = batch_row_ownership_torch(tu, tp, tn, bs, U, I)
    assert stride >= 4 * min(bs, n) and stride & (stride - 1) == 0 and tuple(total.shape) == ((n + bs - 1) // bs, stride)
    bid = (torch.arange(n, device=hip_device) // bs).repeat(3).view(3, n)
    # the kernel gives every valid row a slot; one that occurs once has total == 1 (the step treats it like -1)
    own = torch.where((own >= 0) & (total[bid, own.clamp(min=0).long()] > 1), own, torch.full_like(own, -1))
    assert torch.equal(own >= 0, own_t >= 0)
    sh = own >= 0
    assert torch.equal(total[bid[sh], own[sh].long()], total_t[bid[sh], own_t[sh].long()])
    if n <= 20_000:
        _brute_force_ownership_check(users, pos, neg, bs, U, I, own.cpu().numpy(), total.cpu().numpy())


def _check_row_contributions(users, pos, neg, bs, U, I, chunk, cidx, rows, counts, row_cap, long_row=32, min_c=2):
    """hiprec_batch_row_contrib's contract, batch by batch (numpy)."""
    n = len(users)
    for b in range((n + bs - 1) // bs):
        lo, hi = b * bs, min(n, (b + 1) * bs)
        u, p, q = users[lo:hi], pos[lo:hi], neg[lo:hi]
        ok = (u >= 0) & (u < U) & (p >= 0) & (p < I) & (q >= 0) & (q < I)
        j = np.arange(hi - lo)
        head = np.ones(hi - lo, dtype=bool)
        head[1:] = (p[1:] != p[:-1]) | ~ok[:-1]
        head |= j % chunk == 0
        c = cidx[:, lo:hi]
        assert np.all(c[:, ~ok] == -1), "a triple with an out-of-range id contributes nothing"
        assert np.all(c[1, ok & ~head] == -2), "a positive occurrence inside a run rides with the run's head"
        keys = np.concatenate([u[ok], U + p[ok & head], U + q[ok]])
        where = np.concatenate([c[0, ok], c[1, ok & head], c[2, ok]])
        uniq, inv, cnt = np.unique(keys, return_inverse=True, return_counts=True)
        shared = cnt[inv] >= min_c
        assert np.all(where[~shared] == -1), "a row with one contribution is updated by its contributor"
        n_short, n_long, n_contrib = (int(x) for x in counts[b, :3])
        assert n_contrib == int(shared.sum()) and n_short + n_long == int((cnt >= min_c).sum()) <= row_cap
        recs = np.concatenate([rows[b, :n_short], rows[b, row_cap - n_long:][::-1]]) if n_short + n_long else np.zeros((0, 4), np.int64)
        assert np.all(recs[:n_short, 2] <= long_row) and np.all(recs[n_short:, 2] > long_row)
        order = np.argsort(recs[:, 0])
        recs = recs[order]
        assert np.array_equal(recs[:, 0], uniq[cnt >= min_c]) and np.array_equal(recs[:, 2], cnt[cnt >= min_c])
        # the ranges tile [0, n_contrib) and every contribution of a row has a place of its own inside the row's range
        by_start = recs[np.argsort(recs[:, 1])]
        assert np.array_equal(by_start[:, 1], np.concatenate([[0], np.cumsum(by_start[:, 2])[:-1]]))
        assert np.array_equal(np.sort(where[shared]), np.arange(n_contrib))
        start_of = np.full(len(uniq), -1, dtype=np.int64)
        start_of[cnt >= min_c] = recs[:, 1]
        s0 = start_of[inv[shared]]
        assert np.all((where[shared] >= s0) & (where[shared] < s0 + cnt[inv[shared]]))


@pytest.mark.parametrize("n,bs,U,I,D", [(1000, 128, 50, 30, 64), (3 * 4096 + 77, 4096, 100_000, 2_000, 256),
                                        (2 * 65536, 65536, 1_250_000, 125_000, 128), (40, 64, 9, 5, 128)])
def test_batch_row_contrib_kernel(hip_device, n, bs, U, I, D):
    """hiprec_batch_row_contrib (csrc/ownership.hip): the contribution lists of the owner-pulls step against a numpy
    statement of the contract -- heads of positive runs per chunk, one record and one contiguous range per row with
    several contributions, long rows listed from the end, out-of-range triples left out (and they end a run)."""
    from beta_recsys_amd import _lib
    from beta_recsys_amd.mf import batch_row_contributions

    rng = np.random.default_rng(n)
    users, pos, neg = _zipf_triples(rng, n, U, I, hot=0.3 if n > 1000 else None)
    # every batch sorted by positive item, as the batcher leaves it
    for k in range(0, n, bs):
        o = np.argsort(pos[k:k + bs], kind="stable")
        users[k:k + bs], pos[k:k + bs], neg[k:k + bs] = users[k:k + bs][o], pos[k:k + bs][o], neg[k:k + bs][o]
    users[3], pos[min(21, n - 1)], neg[min(30, n - 1)] = U, -1, I
    chunk = _lib.load().hiprec_mf_pull_chunk(D)
    assert chunk == (8 if D <= 128 else 4)
    tu, tp, tn = (torch.from_numpy(a).cuda() for a in (users, pos, neg))
    cidx, rows, counts, row_cap = batch_row_contributions(tu, tp, tn, bs, U, I, D)
    _check_row_contributions(users, pos, neg, bs, U, I, chunk, cidx.cpu().numpy(), rows.cpu().numpy().astype(np.int64),
                             counts.cpu().numpy(), row_cap)
    # the lazy Adam / RMSprop form: every row of a batch has a record, no contribution is applied in place
    cidx, rows, counts, row_cap = batch_row_contributions(tu, tp, tn, bs, U, I, D, every_row=True)
    assert row_cap == 3 * bs
    _check_row_contributions(users, pos, neg, bs, U, I, chunk, cidx.cpu().numpy(), rows.cpu().numpy().astype(np.int64),
                             counts.cpu().numpy(), row_cap, min_c=1)


@pytest.mark.parametrize("n,bs,I,order", [(3 * 9000 + 77, 9000, 50, "perm"), (2 * 65536, 65536, 125_000, "seed"),
                                          (65536 + 5, 65536, 1_000_000, "seq"), (20_000, 16384, 5000, "seed")])
def test_stage_epoch_grouped_kernel(hip_device, n, bs, I, order):
    """hiprec_stage_epoch_grouped (csrc/ownership.hip; batches beyond the LDS sort): every batch holds exactly the triples
    its slice of the visiting order names (perm[] / the Feistel shuffle of a seed / sequential) and is sorted by positive
    item -- what hiprec_stage_sort_keys + a device sort + hiprec_gather_epoch produced, without the sort.  Out-of-range
    items are grouped at the ends (the step flags them)."""
    from beta_recsys_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(n)
    users, pos, neg = _zipf_triples(rng, n, 1000, I, hot=0.2)
    pos[5], pos[n - 3] = -4, I + 7
    tu, tp, tn = (torch.from_numpy(a).cuda() for a in (users, pos, neg))
    perm = torch.from_numpy(rng.permutation(n)).cuda() if order == "perm" else None
    shuffle, seed = (1, 12345) if order == "seed" else (0, 0)
    st = _lib.stream_ptr(hip_device)
    want = [torch.empty_like(t) for t in (tu, tp, tn)]
    _lib.check(lib.hiprec_gather_epoch(_lib.ptr(tu), _lib.ptr(tp), _lib.ptr(tn), _lib.ptr(perm), shuffle, seed, None, n,
                                       *(_lib.ptr(t) for t in want), st))
    ws = torch.empty(lib.hiprec_stage_grouped_ws_ints(n, bs, I), dtype=torch.int32, device=hip_device)
    assert ws.numel() > 0
    got = [torch.empty_like(t) for t in (tu, tp, tn)]
    _lib.check(lib.hiprec_stage_epoch_grouped(_lib.ptr(tu), _lib.ptr(tp), _lib.ptr(tn), _lib.ptr(perm), shuffle, seed, n,
                                              bs, I, _lib.ptr(ws), *(_lib.ptr(t) for t in got), st))
    gu, gp, gn = (t.cpu().numpy() for t in got)
    wu, wp, wn = (t.cpu().numpy() for t in want)
    for k in range(0, n, bs):
        sl = slice(k, min(n, k + bs))
        key = np.clip(gp[sl], 0, I - 1)
        assert np.all(np.diff(key) >= 0), "a batch is not sorted by positive item"
        a = np.stack([gu[sl], gp[sl], gn[sl]], 1)
        b = np.stack([wu[sl], wp[sl], wn[sl]], 1)
        assert np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])]), "a batch lost or gained a triple"
    assert lib.hiprec_stage_grouped_ws_ints(n, bs, 9_000_000) == 0      # beyond 8 M items: the caller keeps its sort


def test_golden_suite_against_the_ieee_arithmetic_build(hip_device):
    """libhiprec_ieee.so (-DHIPREC_IEEE_DIV: ATen's correctly rounded sqrt and division in the Adam / RMSprop
    denominators, op for op) is built by __graft_entry__.build() next to the product library; the step, trajectory
    and fused-epoch parity tests run against it in a fresh interpreter (HIPREC_LIB selects the library), with the
    trajectory bound enforced on every element."""
    import subprocess
    import sys

    from beta_recsys_amd import _lib

    ieee = os.path.join(os.path.dirname(_lib.LIB_PATH), "libhiprec_ieee.so")
    if IEEE_BUILD:
        pytest.skip("already running against the IEEE build")
    assert os.path.exists(ieee), "libhiprec_ieee.so is missing: run __graft_entry__.build()"
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HIPREC_LIB="libhiprec_ieee.so")
    out = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.join(here, "test_mf_gpu.py"), os.path.join(here, "test_ncf_gpu.py"),
         "-m", "gpu", "-x", "-q", "-k",
         "step_matches_reference or multi_step_trajectory or fused_epoch or full_size_dense_optimizer or ncf_step"],
        env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


@pytest.mark.parametrize("n,bs,U,I,shuffle", [(3 * 9000 + 77, 9000, 500, 300, True), (2 * 65536 + 5, 65536, 200_000, 20_000, True),
                                             (20_000, 10_000, 50, 7, False)])
def test_epoch_grouped_by_positive_item_without_a_sort(hip_device, n, bs, U, I, shuffle):
    """Batches beyond the staging kernel's LDS sort (> 8192 triples): hiprec_gather_epoch + the ownership tables +
    hiprec_group_epoch_by_item (csrc/ownership.hip) replace torch.argsort and the index gathers.  Every batch keeps its
    triples (as a multiset), equal positive items are adjacent, the ownership arrays of the new layout meet their
    contract (brute-force count), out-of-range triples are parked at the batch's end with own = -1, and the Feistel
    shuffle evaluated on the fly equals hiprec_random_permutation's."""
    from beta_recsys_amd import _lib
    from beta_recsys_amd.mf import group_epoch_by_item
    from test_host_logic import _brute_force_ownership_check

    lib, dev = _lib.load(), hip_device
    rng = np.random.default_rng(n)
    users, pos, neg = _zipf_triples(rng, n, U, I)
    bad = rng.integers(0, n, 3)
    pos[bad[0]], users[bad[1]], neg[bad[2]] = I, U + 5, -1          # three triples with an out-of-range id
    tu, tp, tn = (torch.from_numpy(a).to(dev) for a in (users, pos, neg))
    seed = 12345
    ou, op, on = torch.empty_like(tu), torch.empty_like(tp), torch.empty_like(tn)
    _lib.check(lib.hiprec_gather_epoch(_lib.ptr(tu), _lib.ptr(tp), _lib.ptr(tn), None, 1 if shuffle else 0, seed, None, n,
                                       _lib.ptr(ou), _lib.ptr(op), _lib.ptr(on), _lib.stream_ptr(dev)))
    perm = torch.arange(n, device=dev)
    if shuffle:
        _lib.check(lib.hiprec_random_permutation(_lib.ptr(perm), n, seed, _lib.stream_ptr(dev)))
    assert torch.equal(ou, tu[perm]) and torch.equal(op, tp[perm]) and torch.equal(on, tn[perm])
    # the engines' path: sort keys (batch, clamped item) + one device sort + a gather through its permutation
    keys = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.check(lib.hiprec_stage_sort_keys(_lib.ptr(tp), None, 1 if shuffle else 0, seed, n, bs, I, 4, _lib.ptr(keys),
                                          _lib.stream_ptr(dev)))
    assert torch.equal(keys.long(), torch.arange(n, device=dev) // bs * I + tp[perm].clamp(0, I - 1))
    order = torch.sort(keys).indices
    su, sp, sn = torch.empty_like(tu), torch.empty_like(tp), torch.empty_like(tn)
    _lib.check(lib.hiprec_gather_epoch(_lib.ptr(tu), _lib.ptr(tp), _lib.ptr(tn), None, 1 if shuffle else 0, seed,
                                       _lib.ptr(order), n, _lib.ptr(su), _lib.ptr(sp), _lib.ptr(sn), _lib.stream_ptr(dev)))
    assert torch.equal(su, ou[order]) and torch.equal(sp, op[order]) and torch.equal(sn, on[order])
    gu, gp, gn, (own, total, stride) = group_epoch_by_item(ou, op, on, bs, U, I)
    vu, vp, vn = (t.cpu().numpy() for t in (ou, op, on))
    hu, hp_, hn = (t.cpu().numpy() for t in (gu, gp, gn))
    hown = own.cpu().numpy()
    for k in range(0, n, bs):
        a = np.stack([vu[k:k + bs], vp[k:k + bs], vn[k:k + bs]], 1)
        b = np.stack([hu[k:k + bs], hp_[k:k + bs], hn[k:k + bs]], 1)
        assert np.array_equal(a[np.lexsort(a.T)], b[np.lexsort(b.T)]), "a batch lost or gained triples"
        ok = (b[:, 0] >= 0) & (b[:, 0] < U) & (b[:, 1] >= 0) & (b[:, 1] < I) & (b[:, 2] >= 0) & (b[:, 2] < I)
        n_ok = int(ok.sum())
        assert ok[:n_ok].all(), "out-of-range triples must sit at the end of their batch"
        assert (hown[:, k + n_ok:k + len(b)] == -1).all()
        p = b[:n_ok, 1]
        starts = np.flatnonzero(np.r_[True, p[1:] != p[:-1]])
        assert len(starts) == len(np.unique(p)), "equal positive items must be adjacent"
    # the kernel gives every valid row a slot; one that occurs once has total == 1 (the step treats it like -1)
    bid = (torch.arange(n, device=dev) // bs).repeat(3).view(3, n)
    shared = (own >= 0) & (total[bid, own.clamp(min=0).long()] > 1)
    assert bool(((own >= 0) | (total[bid, own.clamp(min=0).long()] >= 0)).all())
    if n <= 30_000:
        norm = torch.where(shared, own, torch.full_like(own, -1))
        _brute_force_ownership_check(hu, hp_, hn, bs, U, I, norm.cpu().numpy(), total.cpu().numpy())
    else:   # vectorised: a slot's total equals the number of occurrences that point at it, per batch
        flat = (bid.long() * stride + own.clamp(min=0).long())[own >= 0]
        cnt = torch.bincount(flat, minlength=total.numel()).view_as(total)
        used = cnt > 0
        assert torch.equal(cnt[used], total[used].long())
