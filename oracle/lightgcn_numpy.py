"""ORACLE (test infrastructure — never imported by the product path).

numpy/scipy fp32 restatement of the beta-recsys LightGCN training step.  Each function cites the
reference lines (relative to /root/reference/beta_rec/).  Pinned against golden vectors captured
from the real reference (oracle/gen_golden.py --lightgcn -> tests/golden/lightgcn_*.npz; checked
by tests/test_oracle_golden_lightgcn.py).

Parameters: {"user_embedding.weight" [U, D], "item_embedding.weight" [I, D]} (models/lightgcn.py:22-23).
The graph is the reference's norm_adj = D^-1 (A + I) over the (U+I) x (U+I) bipartite adjacency
(utils/common_util.py:24-41, data/deprecated_data_base.py:353), given here as scipy CSR.
"""
import numpy as np
import scipy.sparse as sp

from .mf_numpy import F32, opt_step  # noqa: F401

KEYS = ("user_embedding.weight", "item_embedding.weight")


def build_norm_adj(n_users, n_items, users, items):
    """D^-1 (A + I) in CSR, fp32 (deprecated_data_base.py:331-353 + normalized_adj_single)."""
    N = n_users + n_items
    r = np.concatenate([users, items + n_users])
    c = np.concatenate([items + n_users, users])
    adj = sp.coo_matrix((np.ones(len(r), dtype=np.float32), (r, c)), shape=(N, N)).tocsr()
    adj.data[:] = 1.0  # duplicates collapse to one edge
    adj = adj + sp.eye(N, dtype=np.float32, format="csr")
    rowsum = np.asarray(adj.sum(1)).flatten()
    d_inv = np.where(rowsum > 0, 1.0 / rowsum, 0.0)
    out = sp.diags(d_inv).dot(adj).tocsr().astype(np.float32)
    out.sort_indices()
    return out


def apply_edge_dropout(adj_csr, keep_mask, keep_prob):
    """LightGCN.dropout (models/lightgcn.py:27-38): keep the edges where keep_mask, scale 1/keep.
    keep_mask is indexed in coalesced COO (= sorted CSR) order."""
    a = adj_csr.tocoo()
    keep = np.asarray(keep_mask, dtype=bool)
    vals = (a.data[keep] / F32(keep_prob)).astype(F32)
    return sp.csr_matrix((vals, (a.row[keep], a.col[keep])), shape=a.shape)


def propagate(w, adj, n_layers):
    """LightGCN.forward (models/lightgcn.py:46-78): mean over layers of A^l E0."""
    e = np.concatenate([w["user_embedding.weight"], w["item_embedding.weight"]], axis=0).astype(F32)
    acc = e.copy()
    for _ in range(n_layers):
        e = (adj @ e).astype(F32)
        acc += e
    out = (acc / F32(n_layers + 1)).astype(F32)
    U = w["user_embedding.weight"].shape[0]
    return out[:U], out[U:]


def softplus(x):
    x = x.astype(F32)
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, F32(20)), dtype=F32), dtype=F32)).astype(F32)


def lightgcn_grads(w, adj, n_layers, users, pos, neg, decay):
    """LightGCNEngine.train_single_batch without the optimizer (models/lightgcn.py:119-149,
    loss_comput :171-191): returns (loss = mf + reg, dense grads)."""
    users, pos, neg = (np.asarray(a, dtype=np.int64) for a in (users, pos, neg))
    U = w["user_embedding.weight"].shape[0]
    B = F32(len(users))
    ua, ia = propagate(w, adj, n_layers)
    ue, pe, ne = ua[users], ia[pos], ia[neg]
    pos_s = (ue * pe).sum(axis=1, dtype=F32)
    neg_s = (ue * ne).sum(axis=1, dtype=F32)
    x = (neg_s - pos_s).astype(F32)
    mf = F32(softplus(x).mean(dtype=F32))
    u0, p0, n0 = w["user_embedding.weight"][users], w["item_embedding.weight"][pos], w["item_embedding.weight"][neg]
    reg = F32(0.5) * ((u0 ** 2).sum(dtype=F32) + (p0 ** 2).sum(dtype=F32) + (n0 ** 2).sum(dtype=F32)) / B
    reg = F32(reg * F32(decay))
    # backward: d mf / d x = sigmoid(x) / B
    dx = (F32(1) / (F32(1) + np.exp(-x, dtype=F32)) / B).astype(F32)
    N = adj.shape[0]
    d_out = np.zeros((N, ue.shape[1]), dtype=F32)
    np.add.at(d_out, users, dx[:, None] * (ne - pe))
    np.add.at(d_out, pos + U, -dx[:, None] * ue)
    np.add.at(d_out, neg + U, dx[:, None] * ue)
    d_out /= F32(n_layers + 1)
    # out = (E0 + A E0 + ... + A^L E0)/(L+1)  =>  dE0 = sum_l (A^T)^l d_out
    at = adj.T.tocsr()
    g_e, cur = d_out.copy(), d_out
    for _ in range(n_layers):
        cur = (at @ cur).astype(F32)
        g_e += cur
    # L2 on the layer-0 rows: d reg / d row = decay * row / B per occurrence
    c = F32(decay) / B
    np.add.at(g_e, users, c * u0)
    np.add.at(g_e, pos + U, c * p0)
    np.add.at(g_e, neg + U, c * n0)
    grads = {"user_embedding.weight": g_e[:U], "item_embedding.weight": g_e[U:]}
    return float(mf + reg), grads


def lightgcn_predict(w, adj, n_layers, users, items):
    """LightGCN.predict (models/lightgcn.py:80-101): eval mode (no dropout), sigmoid of the dot."""
    ua, ia = propagate(w, adj, n_layers)
    s = (ua[np.asarray(users)] * ia[np.asarray(items)]).sum(axis=1, dtype=F32)
    return (F32(1) / (F32(1) + np.exp(-s, dtype=F32))).astype(F32)


def new_opt_state(w, optimizer):
    st = {"step": 0}
    z = lambda: {k: np.zeros_like(v) for k, v in w.items()}  # noqa: E731
    if optimizer == "adam":
        st["exp_avg"], st["exp_avg_sq"] = z(), z()
    elif optimizer == "rmsprop":
        st["square_avg"] = z()
    return st
