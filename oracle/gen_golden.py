"""ORACLE tooling (test infrastructure): capture golden vectors from the REAL reference.

Runs only in the build container, where /root/reference exists; the reference itself never
travels (no source, no bytecode) — only the small .npz fixtures written to tests/golden/ do.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Imports beta_rec.models.mf / beta_rec.data.data_loaders from /root/reference with the two
in-process shims SURVEY.md §8c lists (a no-op ``tensorboardX`` module; the numpy aliases removed in
numpy 1.24) and drives the reference's own ``MFEngine`` on seeded synthetic inputs.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
KEYS = ("global_bias", "user_emb.weight", "item_emb.weight", "user_bias.weight", "item_bias.weight")


def import_reference():
    sys.dont_write_bytecode = True
    tb = types.ModuleType("tensorboardX")

    class SummaryWriter:  # records scalars so the epoch fixture can pin them
        def __init__(self, *a, **k):
            self.scalars = []

        def add_scalar(self, tag, value, step=None):
            self.scalars.append((tag, float(value), step))

    tb.SummaryWriter = SummaryWriter
    sys.modules["tensorboardX"] = tb
    for name, typ in (("int", int), ("long", int), ("float", float), ("bool", bool)):
        if not hasattr(np, name):
            setattr(np, name, typ)
    sys.path.insert(0, REF)
    from beta_rec.data.data_loaders import PairwiseNegativeDataset, RatingDataset
    from beta_rec.models.mf import MFEngine

    return MFEngine, PairwiseNegativeDataset, RatingDataset


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def make_config(U, I, D, optimizer, loss, lr, B, **extra):
    model = dict(n_users=U, n_items=I, emb_dim=D, device_str="cpu", optimizer=optimizer, lr=lr,
                 batch_size=B, loss=loss)
    model.update(extra)
    return {"model": model, "system": {"run_dir": "/tmp/hiprec_golden_runs"}}


def zipf_items(rng, n, I):
    p = 1.0 / np.arange(1, I + 1)
    p /= p.sum()
    return rng.permutation(I)[rng.choice(I, size=n, p=p)]


def state_np(model, prefix):
    return {f"{prefix}/{k}": v.detach().numpy().copy() for k, v in model.state_dict().items()}


def perturb_biases(model, rng):
    """The reference starts biases at 0; nudge them so the fixtures exercise every term."""
    with torch.no_grad():
        model.user_bias.weight.copy_(torch.from_numpy(
            (rng.standard_normal(model.user_bias.weight.shape) * 0.01).astype(np.float32)))
        model.item_bias.weight.copy_(torch.from_numpy(
            (rng.standard_normal(model.item_bias.weight.shape) * 0.01).astype(np.float32)))
        model.global_bias.fill_(0.05)


def steps_fixture(MFEngine, name, U, I, D, B, optimizer, loss, lr, n_steps, seed, top_reg=None, hot=None):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    cfg = make_config(U, I, D, optimizer, loss, lr, B)
    if top_reg is not None:
        # quirk Q1 (models/mf.py:81-83): `reg` only takes effect when the key ALSO exists at the
        # top level of the config; this fixture pins the L2 term's gradient for that case.
        cfg["reg"] = top_reg
        cfg["model"]["reg"] = top_reg
    eng = quiet(MFEngine, cfg)
    assert eng.reg == (top_reg if top_reg is not None else 0.0)
    perturb_biases(eng.model, rng)
    out = {"meta": np.array([U, I, D, B, n_steps], dtype=np.int64),
           "optimizer": np.array(optimizer), "loss_kind": np.array(loss), "lr": np.array(lr),
           "reg_coef": np.array(0.0 if top_reg is None else top_reg)}
    out.update(state_np(eng.model, "w0"))
    grads_seen = []
    orig_step = eng.optimizer.step

    def capturing_step(*a, **k):
        grads_seen.append({n: (p.grad.detach().numpy().copy() if p.grad is not None
                               else np.zeros(tuple(p.shape), np.float32))
                           for n, p in eng.model.named_parameters()})
        return orig_step(*a, **k)

    eng.optimizer.step = capturing_step
    users = rng.integers(0, U, size=(n_steps, B))
    a_items = np.stack([zipf_items(rng, B, I) for _ in range(n_steps)])
    if hot is not None:
        # a heavy-duplicate batch: ONE item holds the fraction `hot` of the positives (the LDS run-merge path of the
        # gradient kernels)
        a_items[rng.random((n_steps, B)) < hot] = 3
    if loss == "bpr":
        third = rng.integers(0, I, size=(n_steps, B))
    else:
        third = (rng.random((n_steps, B)) < 0.3).astype(np.float32)
    losses, regs = [], []
    for s in range(n_steps):
        batch = (torch.from_numpy(users[s]), torch.from_numpy(a_items[s]),
                 torch.from_numpy(third[s]))
        l, r = eng.train_single_batch(batch)
        losses.append(l)
        regs.append(r)
        out.update(state_np(eng.model, f"w{s + 1}"))
        for k, v in grads_seen[-1].items():
            out[f"g{s + 1}/{k}"] = v
        # optimizer state after this step (exp_avg / exp_avg_sq / square_avg), per parameter
        for pname, p in eng.model.named_parameters():
            pst = eng.optimizer.state.get(p, {})
            for sk, tag in (("exp_avg", "m"), ("exp_avg_sq", "v"), ("square_avg", "v")):
                if sk in pst:
                    out[f"{tag}{s + 1}/{pname}"] = pst[sk].detach().numpy().copy()
    out["users"], out["items_a"], out["third"] = users, a_items, third
    out["losses"] = np.array(losses, dtype=np.float64)
    out["regs"] = np.array(regs, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: losses {losses}")


def kat_fixture(MFEngine):
    """The tiny known-answer vector of SURVEY.md §8c."""
    eng = quiet(MFEngine, make_config(4, 5, 4, "sgd", "bpr", 0.1, 3))
    with torch.no_grad():
        eng.model.user_emb.weight.copy_(((torch.arange(16).reshape(4, 4) % 7) - 3) * 0.1)
        eng.model.item_emb.weight.copy_(((torch.arange(20).reshape(5, 4) % 5) - 2) * 0.2)
        eng.model.user_bias.weight.copy_((torch.arange(4) * 0.01).reshape(4, 1))
        eng.model.item_bias.weight.copy_((-torch.arange(5) * 0.02).reshape(5, 1))
        eng.model.global_bias.fill_(0.05)
    out = state_np(eng.model, "w0")
    u, p, n = torch.tensor([0, 2, 2]), torch.tensor([1, 3, 0]), torch.tensor([4, 1, 3])
    l, r = eng.train_single_batch((u, p, n))
    out.update(state_np(eng.model, "w1"))
    out.update(users=u.numpy(), pos=p.numpy(), neg=n.numpy(), loss=np.array(l), reg=np.array(r))
    np.savez_compressed(os.path.join(OUT, "mf_kat.npz"), **out)
    print("mf_kat:", l, r)


def init_fixture(MFEngine):
    """Initial weights for a given torch seed (models/mf.py:21-30 RNG consumption order)."""
    out = {}
    for tag, (U, I, D, seed) in {"a": (11, 7, 5, 2020), "b": (40, 33, 64, 7)}.items():
        torch.manual_seed(seed)
        eng = quiet(MFEngine, make_config(U, I, D, "sgd", "bpr", 0.1, 3))
        out[f"{tag}/meta"] = np.array([U, I, D, seed], dtype=np.int64)
        out.update(state_np(eng.model, f"{tag}/w"))
    np.savez_compressed(os.path.join(OUT, "mf_init.npz"), **out)
    print("mf_init written")


def predict_fixture(MFEngine):
    rng = np.random.default_rng(5)
    torch.manual_seed(5)
    U, I, D = 53, 47, 64
    eng = quiet(MFEngine, make_config(U, I, D, "sgd", "bpr", 0.1, 3))
    perturb_biases(eng.model, rng)
    users = rng.integers(0, U, size=301)
    items = rng.integers(0, I, size=301)
    scores = eng.model.predict(users, items).numpy()
    out = state_np(eng.model, "w")
    out.update(users=users, items=items, scores=scores)
    np.savez_compressed(os.path.join(OUT, "mf_predict.npz"), **out)
    print("mf_predict written")


def epoch_fixture(MFEngine, PairwiseNegativeDataset, name, optimizer, seed):
    """One train_an_epoch through the reference's own DataLoader(shuffle=True): pins the batch
    composition rule (shuffle per epoch, last batch short), the (loss, reg) sequence, the
    add_scalar values and the final weights (models/mf.py:121-139, data/base_data.py:247-253)."""
    from torch.utils.data import DataLoader

    rng = np.random.default_rng(seed)
    U, I, D, B, N = 31, 29, 8, 16, 103
    torch.manual_seed(seed)
    eng = quiet(MFEngine, make_config(U, I, D, optimizer, "bpr", 0.05, B))
    perturb_biases(eng.model, rng)
    users = rng.integers(0, U, size=N)
    pos = zipf_items(rng, N, I)
    neg = rng.integers(0, I, size=N)
    out = state_np(eng.model, "w0")
    ds = PairwiseNegativeDataset(torch.LongTensor(users), torch.LongTensor(pos),
                                 torch.LongTensor(neg))
    loader = DataLoader(ds, batch_size=B, shuffle=True)
    seen, results = [], []
    orig = eng.train_single_batch

    def recording(batch):
        seen.append(torch.stack([b.clone() for b in batch]).numpy())
        res = orig(batch)
        results.append(res)
        return res

    eng.train_single_batch = recording
    torch.manual_seed(seed + 1)  # the state the test must start the epoch from
    quiet(eng.train_an_epoch, loader, 0)
    out.update(state_np(eng.model, "w1"))
    out.update(users=users, pos=pos, neg=neg,
               meta=np.array([U, I, D, B, N, seed + 1], dtype=np.int64),
               optimizer=np.array(optimizer),
               batch_sizes=np.array([s.shape[1] for s in seen], dtype=np.int64),
               batches=np.concatenate(seen, axis=1),
               losses=np.array([r[0] for r in results]), regs=np.array([r[1] for r in results]),
               scalar_loss=np.array([s[1] for s in eng.writer.scalars if s[0] == "model/loss"]),
               scalar_reg=np.array([s[1] for s in eng.writer.scalars
                                    if s[0] == "model/regularizer"]))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: {len(seen)} batches, sizes {[s.shape[1] for s in seen]}")


def main():
    os.makedirs(OUT, exist_ok=True)
    MFEngine, PairwiseNegativeDataset, _ = import_reference()
    kat_fixture(MFEngine)
    init_fixture(MFEngine)
    predict_fixture(MFEngine)
    for opt, lr in (("sgd", 0.05), ("adam", 0.05), ("rmsprop", 0.01)):
        steps_fixture(MFEngine, f"mf_bpr_{opt}", 97, 61, 64, 37, opt, "bpr", lr, 3, seed=11)
    steps_fixture(MFEngine, "mf_bce_sgd", 97, 61, 64, 37, "sgd", "bce", 0.05, 3, seed=12)
    steps_fixture(MFEngine, "mf_bce_adam", 97, 61, 64, 37, "adam", "bce", 0.05, 3, seed=13)
    # embedding widths that exercise every kernel specialisation (<=64, <=128, <=256, generic)
    steps_fixture(MFEngine, "mf_bpr_sgd_d4", 23, 19, 4, 9, "sgd", "bpr", 0.1, 2, seed=14)
    steps_fixture(MFEngine, "mf_bpr_sgd_d100", 23, 19, 100, 9, "sgd", "bpr", 0.1, 2, seed=15)
    steps_fixture(MFEngine, "mf_bpr_adam_d200", 23, 19, 200, 9, "adam", "bpr", 0.05, 2, seed=16)
    steps_fixture(MFEngine, "mf_bpr_sgd_d300", 23, 19, 300, 9, "sgd", "bpr", 0.1, 2, seed=17)
    steps_fixture(MFEngine, "mf_bpr_sgd_reg", 23, 19, 64, 9, "sgd", "bpr", 0.1, 2, seed=18,
                  top_reg=0.01)
    steps_fixture(MFEngine, "mf_bce_sgd_reg", 23, 19, 64, 9, "sgd", "bce", 0.1, 2, seed=19,
                  top_reg=0.01)
    epoch_fixture(MFEngine, PairwiseNegativeDataset, "mf_epoch_adam", "adam", seed=21)
    epoch_fixture(MFEngine, PairwiseNegativeDataset, "mf_epoch_sgd", "sgd", seed=22)


def main_hot():
    MFEngine, _, _ = import_reference()
    steps_fixture(MFEngine, "mf_bpr_adam_hot", 300, 120, 64, 320, "adam", "bpr", 0.05, 2, seed=31, hot=0.6)
    steps_fixture(MFEngine, "mf_bpr_sgd_hot", 300, 120, 64, 320, "sgd", "bpr", 0.05, 2, seed=32, hot=0.6)


if __name__ == "__main__" and "--hot" in sys.argv:
    main_hot()

if __name__ == "__main__" and not {"--hot", "--ncf", "--lightgcn", "--eval", "--sampler", "--c1", "--pgmf", "--t2v", "--ngcf", "--ncf-dropout"} & set(sys.argv):
    main()


# ---- NCF family (models/ncf.py, gmf.py, mlp.py) ----------------------------------------------------

def ncf_config(U, I, E, L, optimizer, lr, B, model="ncf_end"):
    return {"model": dict(n_users=U, n_items=I, emb_dim=E, dropout=0.0, device_str="cpu",
                          optimizer=optimizer, lr=lr, batch_size=B, model=model,
                          mlp_config={"n_layers": L, "name": "mlp", "save_name": "mlp.model"},
                          gmf_config={"name": "gmf", "save_name": "gmf.model"}),
            "system": {"run_dir": "/tmp/hiprec_golden_runs", "model_save_dir": "/tmp/hiprec_golden_runs"}}


def ncf_steps_fixture(engine_cls, name, kind, U, I, E, L, B, optimizer, lr, n_steps, seed):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    eng = quiet(engine_cls, ncf_config(U, I, E, L, optimizer, lr, B))
    out = {"meta": np.array([U, I, E, L, B, n_steps, seed], dtype=np.int64), "kind": np.array(kind),
           "optimizer": np.array(optimizer), "lr": np.array(lr)}
    out.update(state_np(eng.model, "w0"))  # also pins the init (quirk Q8) for this torch seed
    grads_seen = []
    orig_step = eng.optimizer.step

    def capturing_step(*a, **k):
        grads_seen.append({n: p.grad.detach().numpy().copy() for n, p in eng.model.named_parameters()})
        return orig_step(*a, **k)

    eng.optimizer.step = capturing_step
    users = rng.integers(0, U, size=(n_steps, B))
    items = np.stack([zipf_items(rng, B, I) for _ in range(n_steps)])
    ratings = (rng.random((n_steps, B)) < 0.25).astype(np.float32)
    losses = []
    for s in range(n_steps):
        losses.append(eng.train_single_batch(torch.from_numpy(users[s]), torch.from_numpy(items[s]),
                                             torch.from_numpy(ratings[s])))
        out.update(state_np(eng.model, f"w{s + 1}"))
        for k, v in grads_seen[-1].items():
            out[f"g{s + 1}/{k}"] = v
        for pname, p in eng.model.named_parameters():
            pst = eng.optimizer.state.get(p, {})
            for sk, tag in (("exp_avg", "m"), ("exp_avg_sq", "v"), ("square_avg", "v")):
                if sk in pst:
                    out[f"{tag}{s + 1}/{pname}"] = pst[sk].detach().numpy().copy()
    # scores of the final model on a fixed probe (model.predict contract)
    pu, pi = rng.integers(0, U, 64), rng.integers(0, I, 64)
    out["probe_users"], out["probe_items"] = pu, pi
    out["probe_scores"] = eng.model.predict(pu, pi).numpy()
    out.update(users=users, items=items, ratings=ratings, losses=np.array(losses, dtype=np.float64))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: losses {losses}")


def main_ncf():
    import_reference()
    from beta_rec.models.gmf import GMFEngine
    from beta_rec.models.mlp import MLPEngine
    from beta_rec.models.ncf import NeuMFEngine

    ncf_steps_fixture(NeuMFEngine, "ncf_neumf_adam", "neumf", 53, 41, 8, 3, 37, "adam", 1e-3, 3, seed=31)
    ncf_steps_fixture(NeuMFEngine, "ncf_neumf_sgd_e32", "neumf", 40, 30, 32, 3, 50, "sgd", 0.05, 2, seed=32)
    ncf_steps_fixture(NeuMFEngine, "ncf_neumf_rmsprop_l2", "neumf", 29, 31, 6, 2, 21, "rmsprop", 1e-3, 2, seed=33)
    ncf_steps_fixture(GMFEngine, "ncf_gmf_adam", "gmf", 53, 41, 8, 3, 37, "adam", 1e-3, 3, seed=34)
    ncf_steps_fixture(MLPEngine, "ncf_mlp_adam", "mlp", 53, 41, 8, 3, 37, "adam", 1e-3, 3, seed=35)


if __name__ == "__main__" and "--ncf" in sys.argv:
    main_ncf()


# ---- LightGCN (models/lightgcn.py) ------------------------------------------------------------------

def lightgcn_fixture(name, U, I, D, L, B, n_edges, optimizer, lr, keep, n_steps, seed):
    import scipy.sparse as sp

    import_reference()
    from beta_rec.models.lightgcn import LightGCNEngine
    from beta_rec.utils.common_util import normalized_adj_single

    rng = np.random.default_rng(seed)
    eu = rng.integers(0, U, n_edges)
    ei = zipf_items(rng, n_edges, I)
    # the reference's adjacency (data/deprecated_data_base.py:331-353): R in a dok matrix, A = [[0,R],[R^T,0]],
    # norm_adj = normalized_adj_single(A + I)
    N = U + I
    R = sp.dok_matrix((U, I), dtype=np.float32)
    for a, b in zip(eu, ei):
        R[a, b] = 1.0
    adj = sp.dok_matrix((N, N), dtype=np.float32).tolil()
    adj[:U, U:] = R.tolil()
    adj[U:, :U] = R.tolil().T
    adj = adj.todok()
    norm = quiet(normalized_adj_single, adj + sp.eye(adj.shape[0])).tocoo().astype(np.float32)
    idx = torch.from_numpy(np.vstack((norm.row, norm.col)).astype(np.int64))
    norm_t = torch.sparse_coo_tensor(idx, torch.from_numpy(norm.data), torch.Size(norm.shape))
    torch.manual_seed(seed)
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, layer_size=[D] * L, keep_pro=keep, regs=[1e-5],
                         device_str="cpu", optimizer=optimizer, lr=lr, batch_size=B, norm_adj=norm_t),
           "system": {"run_dir": "/tmp/hiprec_golden_runs"}}
    eng = quiet(LightGCNEngine, cfg)
    out = {"meta": np.array([U, I, D, L, B, n_steps, seed], dtype=np.int64), "optimizer": np.array(optimizer),
           "lr": np.array(lr), "keep": np.array(keep), "decay": np.array(1e-5),
           "edge_users": eu, "edge_items": ei}
    co = norm_t.coalesce()
    out["adj_row"], out["adj_col"] = co.indices()[0].numpy(), co.indices()[1].numpy()
    out["adj_val"] = co.values().numpy()
    out.update(state_np(eng.model, "w0"))
    grads_seen = []
    orig_step = eng.optimizer.step

    def capturing_step(*a, **k):
        grads_seen.append({n: p.grad.detach().numpy().copy() for n, p in eng.model.named_parameters()})
        return orig_step(*a, **k)

    eng.optimizer.step = capturing_step
    users = rng.integers(0, U, size=(n_steps, B))
    pos = np.stack([zipf_items(rng, B, I) for _ in range(n_steps)])
    neg = rng.integers(0, I, size=(n_steps, B))
    losses, masks = [], []
    eng.model.train()
    for s in range(n_steps):
        torch.manual_seed(1000 + s)  # the state the dropout mask of this step is drawn from
        losses.append(eng.train_single_batch((torch.from_numpy(users[s]), torch.from_numpy(pos[s]),
                                              torch.from_numpy(neg[s]))))
        torch.manual_seed(1000 + s)
        masks.append((torch.rand(len(co.values())) + keep).int().bool().numpy())
        out.update(state_np(eng.model, f"w{s + 1}"))
        for k, v in grads_seen[-1].items():
            out[f"g{s + 1}/{k}"] = v
        for pname, p in eng.model.named_parameters():
            pst = eng.optimizer.state.get(p, {})
            for sk, tag in (("exp_avg", "m"), ("exp_avg_sq", "v")):
                if sk in pst:
                    out[f"{tag}{s + 1}/{pname}"] = pst[sk].detach().numpy().copy()
    pu, pi = rng.integers(0, U, 50), rng.integers(0, I, 50)
    out["probe_users"], out["probe_items"] = pu, pi
    out["probe_scores"] = eng.model.predict(pu, pi).numpy()
    out.update(users=users, pos=pos, neg=neg, losses=np.array(losses), masks=np.packbits(np.stack(masks), axis=1),
               nnz=np.array(len(co.values())))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: nnz {len(co.values())} losses {losses}")


if __name__ == "__main__" and "--lightgcn" in sys.argv:
    lightgcn_fixture("lightgcn_adam", 61, 47, 16, 3, 40, 400, "adam", 0.05, 0.6, 3, seed=41)
    lightgcn_fixture("lightgcn_sgd_d64", 37, 29, 64, 2, 24, 250, "sgd", 0.05, 0.6, 2, seed=42)


# ---- ranking evaluation (core/eval_engine.py evaluate -> utils/evaluation.py) ------------------------

RANK_METRICS = ("precision", "recall", "ndcg", "map")


def eval_fixture(name, users, items, ratings, scores, k_list):
    """Run the reference's own evaluate() on one frame and keep inputs + the metric table."""
    import warnings

    import pandas as pd

    import_reference()
    from beta_rec.core.eval_engine import evaluate

    df = pd.DataFrame({"col_user": users, "col_item": items, "col_rating": ratings})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = evaluate(df, scores, list(RANK_METRICS), list(k_list))
    table = np.array([[res[f"{m}@{k}"] for m in RANK_METRICS] for k in k_list], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), users=np.asarray(users, np.int64),
                        items=np.asarray(items, np.int64), ratings=np.asarray(ratings, np.float32),
                        scores=np.asarray(scores, np.float32), k_list=np.asarray(k_list, np.int32),
                        metrics=table)
    print(name, table.round(4).tolist())


def main_eval():
    rng = np.random.default_rng(7)
    # (1) the reference's usual test frame: one held-out positive + 20 sampled negatives per user
    U, C = 40, 21
    users = np.repeat(np.arange(U), C)
    items = np.concatenate([rng.permutation(500)[:C] for _ in range(U)])
    ratings = np.tile(np.r_[1.0, np.zeros(C - 1)], U).astype(np.float32)
    scores = rng.random(U * C).astype(np.float32)
    eval_fixture("eval_leave_one_out", users, items, ratings, scores, [1, 5, 10, 20])
    # (2) ragged, shuffled rows, several positives with rating values 1..5, users without positives,
    #     segments shorter than k; continuous scores (no ties)
    rows = []
    for u in range(60):
        n = int(rng.integers(1, 40))
        its = rng.permutation(300)[:n]
        n_pos = 0 if u % 7 == 0 else int(rng.integers(1, max(2, n // 3 + 1)))
        for j, it in enumerate(its):
            rows.append((u * 3 + 1, it, float(rng.integers(1, 6)) if j < n_pos else 0.0))
    rows = [rows[i] for i in rng.permutation(len(rows))]
    users, items, ratings = (np.array(c) for c in zip(*rows))
    scores = rng.normal(size=len(rows)).astype(np.float32)
    eval_fixture("eval_ragged", users, items, ratings.astype(np.float32), scores, [3, 10, 50])
    # (2b) heavy score ties (quantised scores incl. negatives and +-0), every segment longer than
    #      max(k): pandas' nlargest then takes its stable path (first occurrence wins a tie).  With
    #      k >= len(group) it falls back to an UNSTABLE sort_values (pandas core/methods/selectn.py
    #      "slow method"), whose tie order is a numpy-build detail — not pinned.
    rows = []
    for u in range(40):
        n = int(rng.integers(55, 90))
        its = rng.permutation(300)[:n]
        n_pos = int(rng.integers(1, 12))
        for j, it in enumerate(its):
            rows.append((u, it, float(rng.integers(1, 6)) if j < n_pos else 0.0))
    rows = [rows[i] for i in rng.permutation(len(rows))]
    tu, ti, tr = (np.array(c) for c in zip(*rows))
    tscores = (rng.integers(-3, 4, len(rows)) / 4.0).astype(np.float32)
    tscores[rng.random(len(rows)) < 0.1] = -0.0
    eval_fixture("eval_ties", tu, ti, tr.astype(np.float32), tscores, [3, 10, 50])
    # (3) nothing relevant anywhere -> every metric is 0.0 (evaluation.py:581-582)
    eval_fixture("eval_no_hits", users[:200], items[:200], np.zeros(200, np.float32), scores[:200], [5])
    # (4) fractional ratings below the relevance threshold 1 (evaluation.py:492) and full-catalogue ranking
    U, I = 12, 400
    users = np.repeat(np.arange(U), I)
    items = np.tile(np.arange(I), U)
    ratings = np.where(rng.random(U * I) < 0.03, 1.0, np.where(rng.random(U * I) < 0.05, 0.5, 0.0))
    scores = rng.normal(size=U * I).astype(np.float32)
    eval_fixture("eval_full_catalogue", users, items, ratings.astype(np.float32), scores, [10, 20, 100])


if __name__ == "__main__" and "--eval" in sys.argv:
    main_eval()


# ---- negative sampling / loader builders (data/base_data.py:182-288) ----------------------------------

def main_sampler():
    """Outputs of the reference's own loader builders on a small training frame: what rows and
    tensors they produce (pinned exactly) and which negatives they draw (pinned distributionally)."""
    import random
    import types
    import warnings

    import pandas as pd

    import_reference()
    from beta_rec.data.base_data import BaseData

    rng = np.random.default_rng(23)
    U, I = 30, 40
    rows = set()
    for u in range(U):
        deg = int(rng.integers(1, 30)) if u != 5 else 37   # user 5 has only 3 untouched items
        for i in rng.permutation(I)[:deg]:
            rows.add((u, int(i)))
    rows = sorted(rows)
    rows = [rows[i] for i in rng.permutation(len(rows))]
    users, items = (np.array(c, dtype=np.int64) for c in zip(*rows))
    ratings = rng.integers(1, 6, len(rows)).astype(np.float64)
    train = pd.DataFrame({"col_user": users, "col_item": items, "col_rating": ratings})
    fake = types.SimpleNamespace(train=train, item_id_pool=list(range(I)), n_users=U, n_items=I)
    out = {"meta": np.array([U, I], dtype=np.int64), "train_users": users, "train_items": items,
           "train_ratings": ratings.astype(np.float32)}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        random.seed(2020)
        bpr = quiet(BaseData.instance_bpr_loader, fake, 16, "cpu").dataset
        out["bpr_users"], out["bpr_pos"], out["bpr_neg"] = (t.numpy() for t in (bpr.user_tensor, bpr.pos_item_tensor, bpr.neg_item_tensor))
        bce = quiet(BaseData.instance_bce_loader, fake, 16, "cpu", 3).dataset
        out["bce_users"], out["bce_items"], out["bce_ratings"] = (t.numpy() for t in (bce.user_tensor, bce.item_tensor, bce.target_tensor))
        mul = quiet(BaseData.instance_mul_neg_loader, fake, 16, "cpu", 3).dataset
        out["mul_users"], out["mul_pos"], out["mul_neg"] = (t.numpy() for t in (mul.user_tensor, mul.pos_item_tensor, mul.neg_item_tensor))
        # many independent draws for one row each of a few users: the reference's sampling distribution
        reps = 400
        draws = np.empty((reps, len(rows)), dtype=np.int64)
        for r in range(reps):
            draws[r] = quiet(BaseData.instance_bpr_loader, fake, 16, "cpu").dataset.neg_item_tensor.numpy()
        out["bpr_neg_draws"] = draws.astype(np.int16)
    np.savez_compressed(os.path.join(OUT, "sampler_loaders.npz"), **out)
    print("sampler_loaders:", len(rows), "rows;", {k: v.shape for k, v in out.items() if k.startswith(("bpr", "bce", "mul"))})


if __name__ == "__main__" and "--sampler" in sys.argv:
    main_sampler()


# ---- BASELINE configs[0]: the reference's own CPU-runnable case (configs/mf_default.json on ML-100k) ----

def main_c1():
    """MF at the mf_default.json shape (943 x 1682, emb_dim 64, batch 400, adam lr 0.05, reg 0.001 -- inert,
    quirk Q1): 6 steps of the REAL reference from its own seeded init.  The weights are too big to
    store (1.4 MB per snapshot); the fixture keeps the inputs, the losses and per-tensor checksums."""
    MFEngine, _, _ = import_reference()
    U, I, D, B, n_steps, seed = 943, 1682, 64, 400, 6, 2020
    torch.manual_seed(seed)
    cfg = make_config(U, I, D, "adam", "bpr", 0.05, B, reg=0.001)
    eng = quiet(MFEngine, cfg)
    rng = np.random.default_rng(seed)
    users = rng.integers(0, U, size=(n_steps, B))
    pos = np.stack([zipf_items(rng, B, I) for _ in range(n_steps)])
    neg = rng.integers(0, I, size=(n_steps, B))
    losses, regs = [], []
    for s in range(n_steps):
        l, r = eng.train_single_batch(tuple(torch.from_numpy(a[s]) for a in (users, pos, neg)))
        losses.append(l)
        regs.append(r)
    out = {"meta": np.array([U, I, D, B, n_steps, seed], dtype=np.int64), "users": users, "pos": pos, "neg": neg,
           "losses": np.array(losses, dtype=np.float64), "regs": np.array(regs, dtype=np.float64)}
    for k, v in eng.model.state_dict().items():
        a = v.detach().numpy().astype(np.float64)
        out[f"sum/{k}"] = np.array(a.sum())
        out[f"sumsq/{k}"] = np.array((a * a).sum())
        out[f"head/{k}"] = v.detach().numpy().reshape(-1)[:64].copy()
    np.savez_compressed(os.path.join(OUT, "mf_c1_adam.npz"), **out)
    print("mf_c1_adam: losses", losses)


if __name__ == "__main__" and "--c1" in sys.argv:
    main_c1()


def pgmf_config(U, I, D, B, optimizer, lr, l2_lambda, grad_clip):
    """PairwiseGMFEngine reads flat keys (pairwise_gmf.py:31-33,76-78,107,111) AND, through
    ModelEngine.__init__, config["model"]{device_str,optimizer,lr} + config["system"]["run_dir"]."""
    return {"n_users": U, "n_items": I, "emb_dim": D, "regs": [1e-5], "batch_size": B, "lr": lr,
            "pretrain_l2_lambda": l2_lambda, "grad_clip": grad_clip, "neg_count": 4,
            "model": {"device_str": "cpu", "optimizer": optimizer, "lr": lr},
            "system": {"run_dir": "/tmp/hiprec_golden_runs"}}


def pgmf_fixture(Engine, name, U, I, D, B, optimizer, lr, l2_lambda, grad_clip, n_steps, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    eng = quiet(Engine, pgmf_config(U, I, D, B, optimizer, lr, l2_lambda, grad_clip))
    with torch.no_grad():
        # the reference's init (std 0.01) keeps every score within 1e-4 of zero; scaled fixtures move
        # the scores far enough from 0 for sigmoid and both relu branches to matter
        eng.model.user_memory.weight.mul_(scale)
        eng.model.item_memory.weight.mul_(scale)
    out = {"meta": np.array([U, I, D, B, n_steps, seed], dtype=np.int64), "optimizer": np.array(optimizer),
           "lr": np.array(lr), "l2_lambda": np.array(l2_lambda), "grad_clip": np.array(grad_clip)}
    for k, v in eng.model.state_dict().items():
        out[f"w0/{k}"] = v.detach().numpy().copy()
    seen = []
    orig_step = eng.optimizer.step

    def capturing_step(*a, **k):  # runs after clip_grad_norm_: these are the clipped gradients
        seen.append({n: p.grad.detach().numpy().copy() for n, p in eng.model.named_parameters()})
        return orig_step(*a, **k)

    eng.optimizer.step = capturing_step
    users = rng.integers(0, U, size=(n_steps, B))
    pos = np.stack([zipf_items(rng, B, I) for _ in range(n_steps)])
    neg = rng.integers(0, I, size=(n_steps, B))
    losses = []
    for s in range(n_steps):
        losses.append(eng.train_single_batch((users[s], pos[s], neg[s])))
        for k, v in eng.model.state_dict().items():
            out[f"w{s + 1}/{k}"] = v.detach().numpy().copy()
        for k, v in seen[-1].items():
            out[f"g{s + 1}/{k}"] = v
        for pname, p in eng.model.named_parameters():
            pst = eng.optimizer.state.get(p, {})
            for sk, tag in (("exp_avg", "m"), ("exp_avg_sq", "v"), ("square_avg", "v")):
                if sk in pst:
                    out[f"{tag}{s + 1}/{pname}"] = pst[sk].detach().numpy().copy()
    out["users"], out["pos"], out["neg"] = users, pos, neg
    out["losses"] = np.array(losses, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: losses {losses}")


def main_pgmf():
    """PairwiseGMF (CMN pre-training) fixtures from the real reference engine."""
    import_reference()
    from beta_rec.models.pairwise_gmf import PairwiseGMFEngine

    # cmn_default.json values: adam, l2 1e-4, clip 5.0 (never active at this scale)
    pgmf_fixture(PairwiseGMFEngine, "pgmf_adam", 23, 17, 8, 16, "adam", 1e-3, 1e-4, 5.0, 3, 11, scale=60.0)
    # clip active on every step, width that is not a multiple of 64, sgd
    pgmf_fixture(PairwiseGMFEngine, "pgmf_sgd_clip", 31, 29, 100, 48, "sgd", 0.5, 1e-2, 1e-3, 3, 12, scale=40.0)
    # the reference's own init scale (scores ~1e-4), rmsprop, D 64
    pgmf_fixture(PairwiseGMFEngine, "pgmf_rmsprop_init", 40, 33, 64, 32, "rmsprop", 1e-3, 1e-4, 5.0, 2, 13)
    # seeded initial weights (truncated normal + xavier) for the init-parity test
    out = {}
    for tag, (U, I, D, seed) in {"a": (7, 5, 4, 3), "b": (19, 33, 64, 2020)}.items():
        torch.manual_seed(seed)
        eng = quiet(PairwiseGMFEngine, pgmf_config(U, I, D, 8, "adam", 1e-4, 1e-4, 5.0))
        out[f"{tag}/meta"] = np.array([U, I, D, seed], dtype=np.int64)
        for k, v in eng.model.state_dict().items():
            out[f"{tag}/w/{k}"] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "pgmf_init.npz"), **out)


if __name__ == "__main__" and "--pgmf" in sys.argv:
    main_pgmf()


def t2v_config(U, I, D, B, n_neg, optimizer, lr):
    model = dict(n_users=U, n_items=I, emb_dim=D, n_neg=n_neg, batch_size=B, device_str="cpu",
                 optimizer=optimizer, lr=lr)
    return {"model": model, "system": {"run_dir": "/tmp/hiprec_golden_runs"}}


def t2v_fixture(Engine, name, U, I, D, B, n_neg, optimizer, lr, batch_lens, seed, scale=1.0):
    """batch_lens: real length of each step's batch (a short one pins the / (3 * batch_size) quirk)."""
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    eng = quiet(Engine, t2v_config(U, I, D, B, n_neg, optimizer, lr))
    m = eng.model
    with torch.no_grad():
        for emb in (m.user_emb, m.item_emb1, m.item_emb2):
            emb.weight.mul_(scale)
        m.user_bias.weight.copy_(torch.from_numpy(rng.normal(0, 0.3, (U, 1)).astype(np.float32)))
        m.item_bias.weight.copy_(torch.from_numpy(rng.normal(0, 0.3, (I, 1)).astype(np.float32)))
    n_steps = len(batch_lens)
    out = {"meta": np.array([U, I, D, B, n_neg, n_steps, seed], dtype=np.int64), "optimizer": np.array(optimizer),
           "lr": np.array(lr), "batch_lens": np.array(batch_lens, dtype=np.int64)}
    for k, v in m.state_dict().items():
        out[f"w0/{k}"] = v.detach().numpy().copy()
    names = {id(p): n for n, p in m.named_parameters()}   # captured BEFORE item_emb2 is aliased away
    params = list(m.parameters())
    seen = []
    orig_step = eng.optimizer.step

    def capturing_step(*a, **k):
        seen.append({names[id(p)]: (p.grad.detach().numpy().copy() if p.grad is not None
                                    else np.zeros(tuple(p.shape), np.float32)) for p in params})
        return orig_step(*a, **k)

    eng.optimizer.step = capturing_step
    losses = []
    for s, n in enumerate(batch_lens):
        batch = [rng.integers(0, U, n), zipf_items(rng, n, I), zipf_items(rng, n, I),
                 rng.integers(0, U, (n, n_neg)), rng.integers(0, I, (n, n_neg)), rng.integers(0, I, (n, n_neg))]
        for j, key in enumerate(("pos_u", "pos_i1", "pos_i2", "neg_u", "neg_i1", "neg_i2")):
            out[f"b{s}/{key}"] = batch[j]
        losses.append(eng.train_single_batch(tuple(torch.from_numpy(x) for x in batch)))
        for k, v in m.state_dict().items():
            out[f"w{s + 1}/{k}"] = v.detach().numpy().copy()
        for k, v in seen[-1].items():
            out[f"g{s + 1}/{k}"] = v
        wanted = {"adam": (("exp_avg", "m"), ("exp_avg_sq", "v")), "rmsprop": (("square_avg", "v"),), "sgd": ()}
        for p in params:
            pst = eng.optimizer.state.get(p, {})   # empty for the orphaned item_emb2 (never gets a grad)
            for sk, tag in wanted[optimizer]:
                out[f"{tag}{s + 1}/{names[id(p)]}"] = (pst[sk].detach().numpy().copy() if sk in pst
                                                       else np.zeros(tuple(p.shape), np.float32))
    # predict on the trained model (item_emb2 is item_emb1 by now)
    pu, pi = rng.integers(0, U, 40), rng.integers(0, I, 40)
    out["predict/users"], out["predict/items"] = pu, pi
    out["predict/scores"] = m.predict(pu, pi).numpy()
    out["losses"] = np.array(losses, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: losses {losses}")


def main_t2v():
    """Triple2vec fixtures from the real reference engine + its AliasTable."""
    import_reference()
    from beta_rec.models.triple2vec import Triple2vecEngine
    from beta_rec.utils.alias_table import AliasTable

    t2v_fixture(Triple2vecEngine, "t2v_adam", 21, 17, 8, 12, 3, "adam", 1e-2, [12, 12, 5], 21, scale=60.0)
    t2v_fixture(Triple2vecEngine, "t2v_sgd_d100", 30, 26, 100, 16, 1, "sgd", 0.5, [16, 16], 22, scale=30.0)
    t2v_fixture(Triple2vecEngine, "t2v_rmsprop_init", 25, 31, 64, 8, 5, "rmsprop", 1e-3, [8, 8], 23)
    out = {}
    for tag, (U, I, D, seed) in {"a": (7, 5, 4, 3), "b": (19, 33, 64, 2020)}.items():
        torch.manual_seed(seed)
        eng = quiet(Triple2vecEngine, t2v_config(U, I, D, 8, 2, "adam", 1e-3))
        out[f"{tag}/meta"] = np.array([U, I, D, seed], dtype=np.int64)
        for k, v in eng.model.state_dict().items():
            out[f"{tag}/w/{k}"] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "t2v_init.npz"), **out)
    # AliasTable construction: list and dict frequencies (alias_table.py:24-80)
    rng = np.random.default_rng(5)
    out = {}
    for tag, freq in {"zipf": [int(1000 / (r + 1)) + 1 for r in range(37)],
                      "flat": [3] * 8, "rand": rng.integers(1, 50, 101).tolist()}.items():
        t = quiet(AliasTable, freq)
        out[f"{tag}/freq"] = np.array(freq, dtype=np.int64)
        out[f"{tag}/prob"] = np.asarray(t.prob_arr, dtype=np.float64)
        out[f"{tag}/alias"] = np.asarray(t.alias_arr, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "alias_table.npz"), **out)


if __name__ == "__main__" and "--t2v" in sys.argv:
    main_t2v()


def ngcf_fixture(name, U, I, D, layers, B, n_edges, optimizer, lr, mess_dropout, batch_lens, seed, scale=1.0):
    """NGCFEngine on a small synthetic graph.  The message-dropout masks of every step are captured with
    forward hooks AND re-drawn from the same torch seed the way the product does, to pin that recipe."""
    import scipy.sparse as sp

    import_reference()
    from beta_rec.models.ngcf import NGCFEngine
    from beta_rec.utils.common_util import normalized_adj_single

    rng = np.random.default_rng(seed)
    eu = rng.integers(0, U, n_edges)
    ei = zipf_items(rng, n_edges, I)
    N = U + I
    R = sp.dok_matrix((U, I), dtype=np.float32)
    for a, b in zip(eu, ei):
        R[a, b] = 1.0
    adj = sp.dok_matrix((N, N), dtype=np.float32).tolil()
    adj[:U, U:] = R.tolil()
    adj[U:, :U] = R.tolil().T
    adj = adj.todok()
    norm = quiet(normalized_adj_single, adj + sp.eye(adj.shape[0])).tocoo().astype(np.float32)
    idx = torch.from_numpy(np.vstack((norm.row, norm.col)).astype(np.int64))
    norm_t = torch.sparse_coo_tensor(idx, torch.from_numpy(norm.data), torch.Size(norm.shape))
    torch.manual_seed(seed)
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, layer_size=list(layers), mess_dropout=list(mess_dropout),
                         regs=[1e-5], device_str="cpu", optimizer=optimizer, lr=lr, batch_size=B, norm_adj=norm_t),
           "system": {"run_dir": "/tmp/hiprec_golden_runs"}}
    eng = quiet(NGCFEngine, cfg)
    m = eng.model
    n_steps = len(batch_lens)
    out = {"meta": np.array([U, I, D, len(layers), B, n_steps, seed], dtype=np.int64), "layers": np.array(layers),
           "optimizer": np.array(optimizer), "lr": np.array(lr), "decay": np.array(1e-5),
           "mess_dropout": np.array(mess_dropout, dtype=np.float64), "batch_lens": np.array(batch_lens)}
    co = norm_t.coalesce()
    out["adj_row"], out["adj_col"] = co.indices()[0].numpy(), co.indices()[1].numpy()
    out["adj_val"] = co.values().numpy()
    for k, v in m.state_dict().items():
        out[f"init/{k}"] = v.detach().numpy().copy()          # the seeded init, before any scaling
    with torch.no_grad():
        m.user_embedding.weight.mul_(scale)
        m.item_embedding.weight.mul_(scale)
    for k, v in m.state_dict().items():
        out[f"w0/{k}"] = v.detach().numpy().copy()
    seen, captured = [], []
    orig_step = eng.optimizer.step

    def capturing_step(*a, **k):
        seen.append({n: p.grad.detach().numpy().copy() for n, p in m.named_parameters()})
        return orig_step(*a, **k)

    eng.optimizer.step = capturing_step
    for l, mod in enumerate(m.dropout):
        mod.register_forward_hook(lambda _m, inp, outp, l=l: captured.append((l, inp[0].detach().clone(),
                                                                             outp.detach().clone())))
    m.train()
    losses = []
    for s, n in enumerate(batch_lens):
        users, pos, neg = rng.integers(0, U, n), zipf_items(rng, n, I), rng.integers(0, I, n)
        out[f"b{s}/users"], out[f"b{s}/pos"], out[f"b{s}/neg"] = users, pos, neg
        captured.clear()
        torch.manual_seed(2000 + s)
        loss, reg = eng.train_single_batch(tuple(torch.from_numpy(x) for x in (users, pos, neg)))
        assert reg == 0.0
        losses.append(loss)
        # the product's recipe: same seed, one bernoulli_(1 - p) draw of [N, d] per hop with p > 0, in hop order
        torch.manual_seed(2000 + s)
        for l, x_in, x_out in captured:
            p = mess_dropout[l]
            if p == 0:
                assert torch.equal(x_in, x_out)
                keep = np.ones(tuple(x_in.shape), dtype=bool)
            else:
                keep = torch.empty_like(x_in).bernoulli_(1 - p).bool().numpy()
                ref_keep = (x_out != 0) | (x_in == 0)
                assert np.array_equal(keep | (x_in.numpy() == 0), ref_keep.numpy()), "mask recipe does not replay"
            out[f"mask{s}/{l}"] = np.packbits(keep, axis=1)
        for k, v in m.state_dict().items():
            out[f"w{s + 1}/{k}"] = v.detach().numpy().copy()
        for k, v in seen[-1].items():
            out[f"g{s + 1}/{k}"] = v
        for pname, p_ in m.named_parameters():
            pst = eng.optimizer.state.get(p_, {})
            for sk, tag in (("exp_avg", "m"), ("exp_avg_sq", "v"), ("square_avg", "v")):
                if sk in pst:
                    out[f"{tag}{s + 1}/{pname}"] = pst[sk].detach().numpy().copy()
    m.eval()
    pu, pi = rng.integers(0, U, 50), rng.integers(0, I, 50)
    out["probe_users"], out["probe_items"] = pu, pi
    out["probe_scores"] = m.predict(pu, pi).numpy()
    out["losses"] = np.array(losses, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: nnz {len(co.values())} losses {losses}")


if __name__ == "__main__" and "--ngcf" in sys.argv:
    ngcf_fixture("ngcf_adam", 41, 37, 16, [16, 16, 16], 24, 300, "adam", 0.01, [0.1, 0.1, 0.1], [24, 24, 9], 51,
                 scale=3.0)
    ngcf_fixture("ngcf_sgd_widths", 33, 29, 24, [32, 8], 16, 250, "sgd", 0.5, [0.0, 0.3], [16, 16], 52, scale=3.0)
    ngcf_fixture("ngcf_rmsprop_d64", 50, 45, 64, [64, 64, 64], 32, 400, "rmsprop", 0.001, [0.0, 0.0, 0.0], [32], 53)


def ncf_dropout_fixture(engine_cls, name, kind, U, I, E, L, B, optimizer, lr, dropout, n_steps, seed):
    """NeuMF / MLP with dropout > 0 in the tower (ncf.py:42-45, mlp.py:30-33: Dropout in front of every
    Linear).  The masks of every step are captured with forward hooks AND re-drawn from the same torch
    seed — one bernoulli_(1 - p) of the layer input's shape per Dropout, in layer order — to pin that
    replay recipe (nn.Dropout's own CPU draws)."""
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    cfg = ncf_config(U, I, E, L, optimizer, lr, B)
    cfg["model"]["dropout"] = dropout
    eng = quiet(engine_cls, cfg)
    out = {"meta": np.array([U, I, E, L, B, n_steps, seed], dtype=np.int64), "kind": np.array(kind),
           "optimizer": np.array(optimizer), "lr": np.array(lr), "dropout": np.array(dropout)}
    out.update(state_np(eng.model, "w0"))
    grads_seen, captured = [], []
    orig_step = eng.optimizer.step

    def capturing_step(*a, **k):
        grads_seen.append({n: p.grad.detach().numpy().copy() for n, p in eng.model.named_parameters()})
        return orig_step(*a, **k)

    eng.optimizer.step = capturing_step
    drops = [m for m in eng.model.fc_layers if isinstance(m, torch.nn.Dropout)]
    for l, mod in enumerate(drops):
        mod.register_forward_hook(lambda _m, inp, outp, l=l: captured.append((l, inp[0].detach().clone(),
                                                                             outp.detach().clone())))
    users = rng.integers(0, U, size=(n_steps, B))
    items = np.stack([zipf_items(rng, B, I) for _ in range(n_steps)])
    ratings = (rng.random((n_steps, B)) < 0.25).astype(np.float32)
    losses = []
    eng.model.train()
    for s in range(n_steps):
        captured.clear()
        torch.manual_seed(3000 + s)
        losses.append(eng.train_single_batch(torch.from_numpy(users[s]), torch.from_numpy(items[s]),
                                             torch.from_numpy(ratings[s])))
        assert [c[0] for c in captured] == list(range(len(drops)))
        torch.manual_seed(3000 + s)
        for l, x_in, x_out in captured:
            keep = torch.empty_like(x_in).bernoulli_(1 - dropout).bool().numpy()
            ref_keep = ((x_out != 0) | (x_in == 0)).numpy()
            assert np.array_equal(keep | (x_in.numpy() == 0), ref_keep), "mask recipe does not replay"
            out[f"mask{s}/{l}"] = np.packbits(keep, axis=1)
        out.update(state_np(eng.model, f"w{s + 1}"))
        for k, v in grads_seen[-1].items():
            out[f"g{s + 1}/{k}"] = v
        for pname, p in eng.model.named_parameters():
            pst = eng.optimizer.state.get(p, {})
            for sk, tag in (("exp_avg", "m"), ("exp_avg_sq", "v"), ("square_avg", "v")):
                if sk in pst:
                    out[f"{tag}{s + 1}/{pname}"] = pst[sk].detach().numpy().copy()
    out.update(users=users, items=items, ratings=ratings, losses=np.array(losses, dtype=np.float64))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: losses {losses}")


def main_ncf_dropout():
    import_reference()
    from beta_rec.models.mlp import MLPEngine
    from beta_rec.models.ncf import NeuMFEngine

    ncf_dropout_fixture(NeuMFEngine, "ncf_neumf_dropout", "neumf", 47, 39, 8, 3, 33, "adam", 1e-3, 0.3, 2, seed=36)
    ncf_dropout_fixture(MLPEngine, "ncf_mlp_dropout", "mlp", 47, 39, 8, 2, 33, "sgd", 0.05, 0.5, 2, seed=37)


if __name__ == "__main__" and "--ncf-dropout" in sys.argv:
    main_ncf_dropout()
