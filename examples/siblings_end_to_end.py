"""End-to-end runs of the three sibling models on the device: NGCF, Triple2vec and PairwiseGMF (the CMN
pre-training model) — what ``beta_rec.recommenders.{NGCF,Triple2vec}.train()`` / ``examples/train_cmn.py``
do per run, with every stage on the MI355X.

    python examples/siblings_end_to_end.py [--epochs 4]

Stages (reference file:line -> here):
  * graph            recommenders/ngcf.py:62-67 get_adj_mat + sparse_mx_to_torch_sparse_tensor -> the same
                     D^-1 (A + I) as a torch sparse tensor, converted to CSR on the device once
  * negatives        data/base_data.py:218-253 (BPR triples)          -> beta_recsys_amd.data.sample_negatives
                     models/triple2vec.py:145-166 (alias tables)       -> negative_sampler "device" (hiprec_alias_sample)
  * epochs           models/{ngcf,triple2vec,pairwise_gmf}.py train_an_epoch -> the HIP engines
  * validation       core/eval_engine.py:49-87, 231-274                -> model.predict + beta_recsys_amd.eval.rank_metrics
                     (leave-one-out: 1 held-out positive + sampled negatives per user)
The data is synthetic with planted structure (users and items belong to 8 latent groups and interact
mostly within their group), so that a model that learns ranks a held-out positive above sampled negatives.
Prints one JSON line per model and epoch.
"""
import argparse
import contextlib
import io
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def planted_interactions(n_users, n_items, n, n_groups, seed):
    """Unique (user, item) pairs: 90 % inside the user's group, 10 % anywhere."""
    rng = np.random.default_rng(seed)
    ug, ig = rng.integers(0, n_groups, n_users), rng.integers(0, n_groups, n_items)
    by_group = [np.where(ig == g)[0] for g in range(n_groups)]
    users = rng.integers(0, n_users, int(n * 1.4))
    inside = rng.random(users.size) < 0.9
    items = np.where(inside, [rng.choice(by_group[ug[u]]) for u in users], rng.integers(0, n_items, users.size))
    key = np.unique(users.astype(np.int64) * n_items + items)
    key = key[rng.permutation(key.size)][:n]
    return key // n_items, key % n_items


def norm_adj(n_users, n_items, users, items):
    """D^-1 (A + I) over users + items (data/deprecated_data_base.py:331-353 + normalized_adj_single)."""
    n = n_users + n_items
    r, c = np.concatenate([users, items + n_users]), np.concatenate([items + n_users, users])
    a = sp.coo_matrix((np.ones(len(r), dtype=np.float32), (r, c)), shape=(n, n)).tocsr()
    a.data[:] = 1.0
    a = a + sp.eye(n, dtype=np.float32, format="csr")
    a = sp.diags(1.0 / np.asarray(a.sum(1)).flatten()).dot(a).astype(np.float32).tocoo()
    idx = torch.from_numpy(np.vstack((a.row, a.col)).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(a.data), torch.Size(a.shape))


class FreqTable:
    """The three attributes of utils/alias_table.py::AliasTable the Triple2vec engine reads."""

    def __init__(self, freq):
        freq = np.asarray(freq, dtype=np.float64)
        n = len(freq)
        prob = n * freq / freq.sum()
        alias = np.zeros(n, dtype=np.int64)
        small = [i for i in range(n) if prob[i] < 1.0]
        large = [i for i in range(n) if not prob[i] < 1.0]
        while small and large:
            s, l = small.pop(), large.pop()
            alias[s] = l
            prob[l] -= 1.0 - prob[s]
            (small if prob[l] < 1.0 else large).append(l)
        self.prob_arr, self.alias_arr, self.index2Label = prob, alias, list(range(n))


def validate(model, eval_users, eval_items, ratings, hp):
    scores = model.predict(eval_users, eval_items)
    _, table = hp.eval.rank_metrics(eval_users, ratings, scores, [10])   # columns: precision, recall, ndcg, map
    return float(table[0, 2]), float(table[0, 1])


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=2000)
    ap.add_argument("--items", type=int, default=1500)
    ap.add_argument("--interactions", type=int, default=120_000)
    ap.add_argument("--emb-dim", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--eval-negatives", type=int, default=50)
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X (torch.cuda.is_available() is False)")
    import beta_recsys_amd as hp

    U, I, D, dev = args.users, args.items, args.emb_dim, torch.device("cuda:0")
    users, items = planted_interactions(U, I, args.interactions, 8, seed=1)
    rng = np.random.default_rng(2)
    # leave-one-out: the last interaction of every user (in this order) is held out
    order = np.lexsort((rng.random(users.size), users))
    users, items = users[order], items[order]
    last = np.r_[users[1:] != users[:-1], True]
    tr_u, tr_i, te_u, te_i = users[~last], items[~last], users[last], items[last]
    tu, ti = torch.from_numpy(tr_u).to(dev), torch.from_numpy(tr_i).to(dev)
    # candidates per test user: the held-out positive + sampled untouched items, grouped by user
    k = args.eval_negatives
    all_u, all_i = torch.from_numpy(users).to(dev), torch.from_numpy(items).to(dev)
    neg = hp.data.sample_negatives(all_u, all_i, U, I, k=k, seed=3)[torch.from_numpy(np.flatnonzero(last)).to(dev)]
    ev_u = torch.from_numpy(te_u).to(dev).repeat_interleave(k + 1)
    # the positive goes LAST in its user's block: ties rank by first occurrence, so a constant scorer gets 0
    ev_i = torch.cat([neg, torch.from_numpy(te_i).to(dev)[:, None]], 1).reshape(-1)
    ratings = torch.tensor([0.0] * k + [1.0], device=dev).repeat(len(te_u))
    run_dir = {"run_dir": "/tmp/hiprec_example_runs"}
    history = {}

    def log(name, epoch, loss, ndcg, recall):
        rec = {"model": name, "epoch": epoch, "loss": loss, "ndcg@10": ndcg, "recall@10": recall}
        history.setdefault(name, []).append(rec)
        print(json.dumps(rec), flush=True)

    quiet = lambda: contextlib.redirect_stdout(io.StringIO())  # noqa: E731

    # ---- NGCF: BPR triples with fresh device-side negatives every epoch ---------------------------
    torch.manual_seed(0)
    B = 1024
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, layer_size=[D, D], mess_dropout=[0.1, 0.1], regs=[1e-5],
                         device_str="cuda:0", optimizer="adam", lr=0.005, batch_size=B,
                         norm_adj=norm_adj(U, I, tr_u, tr_i), dropout_rng="device"), "system": run_dir}
    with quiet():
        ngcf = hp.NGCFEngine(cfg)
    for epoch in range(args.epochs):
        negs = hp.data.sample_negatives(tu, ti, U, I, k=1, seed=100 + epoch)[:, 0]
        perm = torch.randperm(tu.numel(), device=dev)
        batches = [(tu[perm[s:s + B]], ti[perm[s:s + B]], negs[perm[s:s + B]]) for s in range(0, tu.numel(), B)]
        with quiet():
            ngcf.train_an_epoch(batches, epoch)
        ngcf.model.eval()
        log("ngcf", epoch, ngcf.writer.scalars[-2][1] / len(batches), *validate(ngcf.model, ev_u, ev_i, ratings, hp))

    # ---- Triple2vec: (user, item, item) triples of co-consumed items, alias-table negatives on the device
    torch.manual_seed(0)
    B = 256
    pairs = np.flatnonzero(tr_u[1:] == tr_u[:-1])            # consecutive interactions of one user
    triples = torch.from_numpy(np.stack([tr_u[pairs], tr_i[pairs], tr_i[pairs + 1]], 1)).to(dev)
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, n_neg=5, batch_size=B, device_str="cuda:0",
                         optimizer="adam", lr=0.005, negative_sampler="device", sampler_seed=7), "system": run_dir}
    with quiet():
        t2v = hp.Triple2vecEngine(cfg)
    t2v.data = type("Data", (), {"user_sampler": FreqTable(np.bincount(tr_u, minlength=U) + 1.0),
                                 "item_sampler": FreqTable(np.bincount(tr_i, minlength=I) + 1.0)})()
    for epoch in range(args.epochs):
        shuffled = triples[torch.randperm(triples.shape[0], device=dev)]
        loader = [shuffled[s:s + B] for s in range(0, shuffled.shape[0], B)]
        with quiet():
            t2v.train_an_epoch(loader, epoch)
        log("triple2vec", epoch, t2v.writer.scalars[-1][1] / len(loader), *validate(t2v.model, ev_u, ev_i, ratings, hp))

    # ---- PairwiseGMF: [B, 3] blocks as cmn_train_loader yields them; scored with its forward -----
    torch.manual_seed(0)
    B = 1024
    cfg = {"n_users": U, "n_items": I, "emb_dim": D, "regs": [1e-5], "batch_size": B, "lr": 0.005,
           "pretrain_l2_lambda": 1e-4, "grad_clip": 5.0, "neg_count": 4,
           "model": {"device_str": "cuda:0", "optimizer": "adam", "lr": 0.005}, "system": run_dir}
    with quiet():
        gmf = hp.PairwiseGMFEngine(cfg)

    class Scorer:  # PairwiseGMF.predict is a stub in the reference; its positive-branch score ranks
        def predict(self, us, its):
            return gmf.model(us, its, its)[0].reshape(-1)

    for epoch in range(args.epochs):
        negs = hp.data.sample_negatives(tu, ti, U, I, k=4, seed=200 + epoch)
        rows = torch.stack([tu.repeat_interleave(4), ti.repeat_interleave(4), negs.reshape(-1)], 1)
        rows = rows[torch.randperm(rows.shape[0], device=dev)]
        loader = [rows[s:s + B] for s in range(0, rows.shape[0], B)]
        with quiet():
            gmf.train_an_epoch(loader, epoch)
        log("pairwise_gmf", epoch, gmf.writer.scalars[-1][1] / len(loader), *validate(Scorer(), ev_u, ev_i, ratings, hp))
    return history


if __name__ == "__main__":
    main()
