"""GPU parity tests of Triple2vec: loss + backward, predict, the full step and the
device alias sampler vs golden vectors from the real reference's Triple2vecEngine / AliasTable, and vs
the numpy oracle at the reference's default shape (emb_dim 64, n_neg 5) on ML-1M-sized tables."""
import contextlib
import io

import numpy as np
import pytest
import torch

from helpers import REL, assert_sgd_exact  # noqa: F401  (north_star: 1e-5 relative)
from helpers import assert_scalar_close, assert_step_close, assert_tensor_close, load_golden
from oracle import triple2vec_numpy as onp
from test_oracle_golden_t2v import CASES, KEYS, bias_floor, t2v_band, t2v_batch, t2v_opt_state, t2v_params

pytestmark = pytest.mark.gpu


def make_engine(U, I, D, B, n_neg, optimizer="adam", lr=1e-3, **extra):
    import beta_recsys_amd as hp

    model = dict(n_users=U, n_items=I, emb_dim=D, n_neg=n_neg, batch_size=B, device_str="cuda:0",
                 optimizer=optimizer, lr=lr)
    model.update(extra)
    with contextlib.redirect_stdout(io.StringIO()):
        return hp.Triple2vecEngine({"model": model, "system": {"run_dir": "/tmp/hiprec_test_runs"}})


def load_weights(eng, w):
    eng.model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})


def get_weights(eng):
    return {k: v.detach().cpu().numpy() for k, v in eng.model.state_dict().items()}


def load_opt_state(eng, st):
    eng.load_optimizer_state(st["step"], st.get("exp_avg"), st.get("exp_avg_sq", st.get("square_avg")))


def aliased(w):
    """What the reference's state_dict shows once item_emb2 IS item_emb1."""
    w = {k: v.copy() for k, v in w.items()}
    w["item_emb2.weight"] = w["item_emb1.weight"].copy()
    return w


@pytest.mark.parametrize("case", CASES)
def test_step_matches_reference(hip_device, case):
    """Each step from the reference's own weights and optimizer state."""
    g = load_golden(case)
    U, I, D, B, n_neg, n_steps, seed = (int(x) for x in g["meta"])
    opt, lr = str(g["optimizer"]), float(g["lr"])
    scale = 1.0 / (3 * B)
    torch.manual_seed(seed)
    eng = make_engine(U, I, D, B, n_neg, opt, lr)
    for s in range(n_steps):
        batch = t2v_batch(g, s)
        w0 = t2v_params(g, f"w{s}")
        st0 = t2v_opt_state(g, s, opt)
        eng.model._alias()
        load_weights(eng, aliased(w0))
        load_opt_state(eng, st0)
        loss, grads = eng.backward_only(batch)
        assert_scalar_close(loss, g["losses"][s], what=f"loss step {s}")
        g_ref = t2v_params(g, f"g{s + 1}")
        for k in KEYS:
            assert_tensor_close(grads[k].cpu().numpy(), g_ref[k], what=f"grad {k} step {s}",
                                scale_floor=bias_floor(k, scale))
        load_opt_state(eng, st0)
        loss = eng.train_single_batch(batch)
        assert_scalar_close(loss, g["losses"][s], what=f"loss (step) {s}")
        band = t2v_band(w0, st0, g_ref, opt, lr, scale)
        band["item_emb2.weight"] = band["item_emb1.weight"]
        w_prev = aliased(w0)
        w1 = get_weights(eng)
        assert np.array_equal(w1["item_emb2.weight"], w1["item_emb1.weight"])
        for k in KEYS:
            assert_step_close(w_prev[k], w1[k], g[f"w{s + 1}/{k}"], band[k], what=f"weights {k} step {s}")
        nxt = t2v_opt_state(g, s + 1, opt)
        for name, buf in (("exp_avg", eng.optimizer.exp_avg), ("exp_avg_sq", eng.optimizer.exp_avg_sq)):
            if buf is None:
                continue
            ref_name = name if opt == "adam" else "square_avg"
            got = {k: v.cpu().numpy() for k, v in eng.model.views(buf).items()}
            for k in KEYS:
                floor = bias_floor(k, scale)
                # the gradient itself is held to 2e-5 here (see below: sums of B * n_neg terms per row); exp_avg is
                # linear in it, the second moments quadratic
                assert_tensor_close(got[k], nxt[ref_name][k], 2e-5 if name == "exp_avg" else 4e-5, f"{name} {k} step {s}",
                                    scale_floor=floor if name == "exp_avg" else floor ** 2)
        assert float(eng._g_flat.abs().max()) == 0.0
    # predict on the reference's final weights
    load_weights(eng, t2v_params(g, f"w{n_steps}"))
    scores = eng.model.predict(g["predict/users"], g["predict/items"])
    assert_tensor_close(scores.cpu().numpy(), g["predict/scores"], what="predict")


def test_first_forward_aliases_item_emb2_like_the_reference(hip_device):
    """From a fresh engine (distinct item_emb2), the first training step already treats item_emb2 as
    item_emb1 (triple2vec.py:38-39) and the trajectory follows the reference's losses."""
    g = load_golden("t2v_sgd_d100")
    U, I, D, B, n_neg, n_steps, _ = (int(x) for x in g["meta"])
    eng = make_engine(U, I, D, B, n_neg, "sgd", float(g["lr"]))
    load_weights(eng, t2v_params(g, "w0"))
    assert not eng.model.shared_items
    for s in range(n_steps):
        loss = eng.train_single_batch(t2v_batch(g, s))
        assert_scalar_close(loss, g["losses"][s], REL, what=f"loss step {s}")
    assert eng.model.shared_items
    w = get_weights(eng)
    # plain SGD has no conditioning problem: every element within 1e-5 of the trajectory's total update (+ 4 ulp)
    assert_sgd_exact(w, t2v_params(g, f"w{n_steps}"), t2v_params(g, "w0"), "final weights")
    # Triple2vec.forward returns the batch loss without touching the weights
    loss = eng.model(t2v_batch(g, 0))
    assert loss.dim() == 0 and loss.device.type == "cuda"
    assert_scalar_close(float(loss), onp.t2v_grads(w, t2v_batch(g, 0), B)[0], what="forward loss")
    assert all(np.array_equal(get_weights(eng)[k], w[k]) for k in KEYS)


@pytest.mark.parametrize("dim,n_neg,optimizer,independent", [(64, 5, "adam", False), (128, 2, "sgd", False),
                                                             (256, 1, "rmsprop", False), (64, 3, "sgd", True)])
def test_default_shape_vs_oracle(hip_device, dim, n_neg, optimizer, independent):
    """triple2vec_default.json shape (emb_dim 64, n_neg 5, batch 256 -> here 2048 to fill the chip) on
    6040 x 3706 tables with Zipf items: loss, gradients and two full steps vs the oracle.
    ``independent``: n_neg = 0 in the config keeps item_emb2 a table of its own (use_bias falsy)."""
    U, I, B = 6040, 3706, 2048
    rng = np.random.default_rng(dim + n_neg)
    torch.manual_seed(dim)
    lr = 0.5 if optimizer == "sgd" else 1e-3
    eng = make_engine(U, I, dim, B, 0 if independent else n_neg, optimizer, lr)
    w = get_weights(eng)
    for k in KEYS[:3]:
        w[k] *= 30.0
    w["user_bias.weight"] = rng.normal(0, 0.3, (U, 1)).astype(np.float32)
    w["item_bias.weight"] = rng.normal(0, 0.3, (I, 1)).astype(np.float32)
    shared = not independent
    if shared:
        eng.model._alias()
        w = aliased(w)
    load_weights(eng, w)
    perm = rng.permutation(I)
    zipf = lambda shape: perm[np.minimum((rng.zipf(1.2, shape) - 1), I - 1)]  # noqa: E731
    st = onp.new_opt_state(w, optimizer)
    scale = 1.0 / (3 * B)
    for step, n in enumerate((B, B - 77)):
        batch = (rng.integers(0, U, n), zipf(n), zipf(n), rng.integers(0, U, (n, n_neg)), zipf((n, n_neg)),
                 zipf((n, n_neg)))
        # the hottest Zipf row receives ~4000 terms here: compare with the order-independent (fp64) sum,
        # at 2e-5 of the tensor's scale (fp32 accumulation ORDER noise ~ sqrt(N) eps on that row)
        loss_o, g_o = onp.t2v_grads(w, batch, B, shared, acc=np.float64)
        if step == 0:
            loss, grads = eng.backward_only(batch)
            eng.load_optimizer_state(0)
            assert_scalar_close(loss, loss_o, what="loss")
            for k in KEYS:
                assert_tensor_close(grads[k].cpu().numpy(), g_o[k], 2e-5, what=f"grad {k}",
                                    scale_floor=bias_floor(k, scale))
            if shared:
                assert not grads["item_emb2.weight"].any()
        w_prev = {k: v.copy() for k, v in w.items()}
        st_prev = {k: ({kk: vv.copy() for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in st.items()}
        loss_o = onp.t2v_train_step(w, st, batch, B, optimizer, lr, shared, acc=np.float64)
        loss = eng.train_single_batch(batch)
        assert_scalar_close(loss, loss_o, what=f"loss step {step}")
        band = t2v_band(w_prev, st_prev, g_o, optimizer, lr, scale, rel=2e-5)
        if shared:
            band["item_emb2.weight"] = band["item_emb1.weight"]
        else:  # t2v_band pins the item_emb2 gradient (right for the aliased case only)
            band["item_emb2.weight"] = band["item_emb2.weight"] + band["item_emb1.weight"].max()
        got = get_weights(eng)
        for k in KEYS:
            assert_step_close(w_prev[k], got[k], w[k], band[k], 2e-5, what=f"weights {k} step {step}")
        load_weights(eng, w)


def test_device_alias_sampler(hip_device):
    """hiprec_alias_sample equals its numpy restatement bit for bit, follows the table's frequencies at
    scale, maps through index2Label, and is a pure function of (seed, element)."""
    from beta_recsys_amd.triple2vec import DeviceAliasTable

    g = load_golden("alias_table")

    class Table:  # the three attributes of utils/alias_table.py the engine reads
        def __init__(self, tag, labels=None):
            self.prob_arr, self.alias_arr = g[f"{tag}/prob"], g[f"{tag}/alias"]
            self.index2Label = list(range(len(self.prob_arr))) if labels is None else labels

    for tag in ("zipf", "flat", "rand"):
        t = DeviceAliasTable(Table(tag), hip_device)
        assert t.labels is None
        got = t.sample(7, 1000, seed=42 + len(tag)).cpu().numpy()
        assert got.shape == (1000, 7)
        assert np.array_equal(got.ravel(), onp.alias_sample(g[f"{tag}/prob"], g[f"{tag}/alias"], None, 42 + len(tag), 7000))
    freq = g["rand/freq"].astype(np.float64)
    t = DeviceAliasTable(Table("rand"), hip_device)
    n = 4_000_000
    counts = torch.bincount(t.sample(4, n // 4, seed=1).reshape(-1), minlength=len(freq)).cpu().numpy()
    expected = n * freq / freq.sum()
    assert np.all(np.abs(counts - expected) < 5 * np.sqrt(expected) + 5)
    labels = [1000 - 3 * i for i in range(len(freq))]
    tl = DeviceAliasTable(Table("rand", labels), hip_device)
    a, b = t.sample(5, 200, seed=9).cpu().numpy(), tl.sample(5, 200, seed=9).cpu().numpy()
    assert np.array_equal(b, np.asarray(labels)[a])
    assert np.array_equal(a, t.sample(5, 200, seed=9).cpu().numpy())
    assert not np.array_equal(a, t.sample(5, 200, seed=10).cpu().numpy())


class FakeAliasTable:
    """AliasTable.sample's contract (count per draw, obj_num draws -> list of lists) on a fixed stream."""

    def __init__(self, prob, alias, seed):
        self.prob_arr, self.alias_arr = prob, alias
        self.index2Label = list(range(len(prob)))
        self.rng = np.random.default_rng(seed)
        self.log = []

    def sample(self, count, obj_num=1, no_repeat=False):
        out = self.rng.integers(0, len(self.prob_arr), (obj_num, count))
        self.log.append(out)
        return out.tolist()


@pytest.mark.parametrize("sampler", ["host", "device"])
def test_train_an_epoch(hip_device, sampler):
    """triple2vec.py:126-169: [B,3] blocks from the loader, three negative draws per batch from
    engine.data's samplers (in the order user, item, item), last loss printed, epoch sum logged."""
    U, I, D, B, n_neg = 60, 45, 16, 32, 3
    g = load_golden("alias_table")
    rng = np.random.default_rng(8)
    triples = np.stack([rng.integers(0, U, 71), rng.integers(0, I, 71), rng.integers(0, I, 71)], 1)
    loader = torch.utils.data.DataLoader(torch.from_numpy(triples).cuda(), batch_size=B, shuffle=False)
    eng = make_engine(U, I, D, B, n_neg, "adam", 1e-2, negative_sampler=sampler, sampler_seed=5)

    class Data:
        user_sampler = FakeAliasTable(g["rand/prob"][:U], np.minimum(g["rand/alias"][:U], U - 1), 1)
        item_sampler = FakeAliasTable(g["zipf/prob"], g["zipf/alias"], 2)

    eng.data = Data()
    w = get_weights(eng)
    for k in KEYS[:3]:
        w[k] *= 50.0
    eng.model._alias()
    w = aliased(w)
    load_weights(eng, w)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        eng.train_an_epoch(loader, 4)
    tag, total, epoch = eng.writer.scalars[-1]
    assert (tag, epoch) == ("model/loss", 4)
    printed = float(out.getvalue().strip().rsplit("Loss ", 1)[1])
    assert np.isfinite(total) and np.isfinite(printed) and total > printed > 0
    if sampler == "device":
        assert not Data.user_sampler.log and not Data.item_sampler.log
        # every draw is a valid id of its table, and a second epoch draws different negatives
        neg = eng._negatives(32, 4, 0)
        assert [tuple(x.shape) for x in neg] == [(32, n_neg)] * 3
        assert int(neg[0].max()) < U and int(neg[1].max()) < len(g["zipf/prob"]) and int(neg[0].min()) >= 0
        assert not torch.equal(neg[1], neg[2]) and not torch.equal(neg[0], eng._negatives(32, 5, 0)[0])
        return
    # host sampler: replay the same negatives through the oracle
    lens = [32, 32, 7]
    assert [x.shape for x in Data.user_sampler.log] == [(n, n_neg) for n in lens]
    assert [x.shape for x in Data.item_sampler.log] == [(n, n_neg) for n in lens for _ in range(2)]
    st = onp.new_opt_state(w, "adam")
    w_start = {k: v.copy() for k, v in w.items()}
    ref, start, np_batches = [], 0, []
    for b, n in enumerate(lens):
        blk = triples[start:start + n]
        start += n
        batch = (blk[:, 0], blk[:, 1], blk[:, 2], Data.user_sampler.log[b], Data.item_sampler.log[2 * b],
                 Data.item_sampler.log[2 * b + 1])
        np_batches.append(batch)
        ref.append(onp.t2v_train_step(w, st, batch, B, "adam", 1e-2))
    assert_scalar_close(total, sum(ref), REL, "epoch loss sum")
    assert_scalar_close(printed, ref[-1], REL, "printed last loss")
    # every element inside the legal-trajectory envelope (helpers.oracle_trajectory)
    from helpers import assert_on_trajectory, oracle_trajectory

    def t2v_step(wc, gr, state):
        onp.opt_step(wc, gr, state, "adam", 1e-2)
        wc["item_emb2.weight"][...] = wc["item_emb1.weight"]   # the aliased table follows (triple2vec.py:38-39)

    w_ref, env, upd = oracle_trajectory(w_start, np_batches, lambda wc, b: onp.t2v_grads(wc, b, B)[1], t2v_step,
                                        lambda wc: onp.new_opt_state(wc, "adam"))
    assert all(np.array_equal(w_ref[k], w[k]) for k in KEYS)
    assert_on_trajectory(get_weights(eng), w_ref, env, upd, "Triple2vec epoch")


def test_bad_indices_and_batches(hip_device):
    eng = make_engine(12, 9, 8, 4, 2)
    ok = ([0, 1], [1, 2], [3, 4], [[1, 2], [3, 4]], [[1, 2], [3, 4]], [[5, 6], [7, 8]])
    for pos, bad in ((0, [0, 12]), (1, [9, 1]), (2, [0, -1]), (3, [[1, 12], [0, 0]]), (4, [[1, 2], [9, 0]]),
                     (5, [[1, 2], [-3, 0]])):
        batch = list(ok)
        batch[pos] = bad
        with pytest.raises(IndexError):
            eng.train_single_batch(tuple(batch))
    with pytest.raises(ValueError):
        eng.train_single_batch(([], [], [], [], [], []))
    with pytest.raises(ValueError):
        eng.train_single_batch(([0, 1], [1], [3, 4], ok[3], ok[4], ok[5]))
    with pytest.raises(ValueError):
        eng.train_single_batch(ok[:5])
    with pytest.raises(IndexError):
        eng.model.predict([0, 12], [0, 0])
    assert np.isfinite(eng.train_single_batch(ok))
