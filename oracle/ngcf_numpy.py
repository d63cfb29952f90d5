"""ORACLE (test infrastructure — never imported by the product path).

numpy/scipy fp32 restatement of the beta-recsys NGCF training step.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference lines (relative to /root/reference/beta_rec/):
    models/ngcf.py:12-46     parameters (GC_weights / Bi_weights Linear per hop, two embeddings) and init
    models/ngcf.py:48-80     forward: side = A ego; ego' = dropout(lrelu(GC(side)) + lrelu(Bi(ego * side)));
                             all = cat(ego0, normalize(ego'_1), ..., normalize(ego'_L))
    models/ngcf.py:82-100    predict
    models/ngcf.py:118-149   train_single_batch
    models/ngcf.py:172-199   bpr_loss (regulariser divided by the CONFIGURED batch size)
Pinned against golden vectors captured from the real reference by ``oracle/gen_golden.py --ngcf``
(tests/golden/ngcf_*.npz); see tests/test_oracle_golden_ngcf.py.

Parameters are a dict with the reference's state_dict keys, in its order:
    GC_weights.{l}.weight [d_{l+1}, d_l], GC_weights.{l}.bias [d_{l+1}] (l = 0..L-1), Bi_weights.{l}.*,
    user_embedding.weight [U, d_0], item_embedding.weight [I, d_0]
The graph is the reference's norm_adj as scipy CSR (see oracle/lightgcn_numpy.build_norm_adj).
``masks``: per hop an [N, d_{l+1}] boolean keep mask of the message dropout (None = no dropout: eval
mode or p == 0); kept entries are scaled by 1 / (1 - p).

``dt``: arithmetic type.  fp32 (default) restates the reference; fp64 gives the exact gradient the
tests measure every fp32 evaluation against: the last hop's row normalisation cancels heavily when
its activations are small (the reference's own fp32 gradient is then up to ~1e-4 of the tensor scale
away from the exact one — tests/test_oracle_golden_ngcf.py shows it on ngcf_rmsprop_d64).
"""
import numpy as np

from .mf_numpy import F32, new_opt_state, opt_step  # noqa: F401

SLOPE = 0.01   # F.leaky_relu default negative_slope
EPS = 1e-12    # F.normalize default eps


def _sigmoid(x):
    return 1 / (1 + np.exp(-x))


def _logsigmoid(x):
    """min(x, 0) - log1p(exp(-|x|)), as ATen computes it."""
    return np.minimum(x, 0) - np.log1p(np.exp(-np.abs(x)))


def keys(n_layers):
    out = []
    for fam in ("GC_weights", "Bi_weights"):
        for l in range(n_layers):
            out += [f"{fam}.{l}.weight", f"{fam}.{l}.bias"]
    return tuple(out + ["user_embedding.weight", "item_embedding.weight"])


def n_layers_of(w):
    return sum(1 for k in w if k.startswith("GC_weights.") and k.endswith(".weight"))


def lrelu(x):
    return np.where(x > 0, x, x * x.dtype.type(SLOPE)).astype(x.dtype)


def ngcf_forward(w, adj, masks=None, drop=None, dt=F32):
    """NGCF.forward (ngcf.py:48-80): (all_embeddings [N, sum d], cache for the backward)."""
    L = n_layers_of(w)
    w = {k: v.astype(dt) for k, v in w.items()}
    adj = adj.astype(dt)
    F32 = dt  # noqa: N806  (every cast below follows the chosen arithmetic type)
    ego = np.concatenate([w["user_embedding.weight"], w["item_embedding.weight"]], axis=0).astype(F32)
    outs, cache = [ego], []
    for l in range(L):
        side = (adj @ ego).astype(F32)
        sum_pre = (side @ w[f"GC_weights.{l}.weight"].T + w[f"GC_weights.{l}.bias"]).astype(F32)
        bi_in = (ego * side).astype(F32)
        bi_pre = (bi_in @ w[f"Bi_weights.{l}.weight"].T + w[f"Bi_weights.{l}.bias"]).astype(F32)
        act = lrelu(sum_pre) + lrelu(bi_pre)
        if masks is not None and masks[l] is not None:
            scale = (masks[l].astype(F32) / F32(1.0 - drop[l])).astype(F32)
        else:
            scale = None
        nxt = act if scale is None else (act * scale).astype(F32)
        nrm = np.sqrt((nxt * nxt).sum(1, dtype=F32), dtype=F32)
        y = (nxt / np.maximum(nrm, F32(EPS))[:, None]).astype(F32)
        cache.append(dict(ego=ego, side=side, sum_pre=sum_pre, bi_in=bi_in, bi_pre=bi_pre, scale=scale,
                          nrm=nrm, y=y))
        outs.append(y)
        ego = nxt
    return np.concatenate(outs, axis=1), cache


def ngcf_grads(w, adj, users, pos, neg, decay, batch_size, masks=None, drop=None, dt=F32, terms=None):
    """zero_grad + forward + bpr_loss + backward of train_single_batch: (loss, grads).  ``terms`` (a dict, optional)
    receives, per bias tensor, the largest column sum of the ABSOLUTE values its gradient adds up: a bias gradient is a
    cancelling sum over all N nodes, and the rounding error of any summation order is relative to that magnitude, not
    to the cancelled result (tests/helpers.py::grad_scale_floor makes the same point for MF's biases)."""
    L = n_layers_of(w)
    U = w["user_embedding.weight"].shape[0]
    F32 = dt  # noqa: N806
    w = {k: v.astype(dt) for k, v in w.items()}
    adj = adj.astype(dt)
    allv, cache = ngcf_forward(w, adj, masks, drop, dt)
    users, pos, neg = (np.asarray(x, dtype=np.int64) for x in (users, pos, neg))
    u, p, n = allv[users], allv[U + pos], allv[U + neg]
    B = F32(len(users))
    x = (u * p).sum(1, dtype=F32) - (u * n).sum(1, dtype=F32)
    reg = (F32(0.5) * (u * u).sum(dtype=F32) + F32(0.5) * (p * p).sum(dtype=F32)
           + F32(0.5) * (n * n).sum(dtype=F32)) / F32(batch_size)
    loss = -_logsigmoid(x).mean(dtype=F32) + F32(decay) * reg
    dx = (-_sigmoid(-x) / B).astype(F32)
    c = F32(decay) / F32(batch_size)
    d_all = np.zeros_like(allv)
    np.add.at(d_all, users, dx[:, None] * (p - n) + c * u)
    np.add.at(d_all, U + pos, dx[:, None] * u + c * p)
    np.add.at(d_all, U + neg, -dx[:, None] * u + c * n)

    g = {}
    widths = [w["user_embedding.weight"].shape[1]] + [w[f"GC_weights.{l}.weight"].shape[0] for l in range(L)]
    offs = np.concatenate([[0], np.cumsum(widths)])
    d_next = None
    adj_t = adj.T.tocsr()
    for l in range(L - 1, -1, -1):
        cch = cache[l]
        d_y = d_all[:, offs[l + 1]:offs[l + 2]]
        nrm, y = cch["nrm"], cch["y"]
        big = nrm >= F32(EPS)
        proj = (y * d_y).sum(1, dtype=F32)[:, None]
        d_x = np.where(big[:, None], (d_y - y * proj) / np.maximum(nrm, F32(EPS))[:, None], d_y / F32(EPS)).astype(F32)
        if d_next is not None:
            d_x = (d_x + d_next).astype(F32)
        d_act = d_x if cch["scale"] is None else (d_x * cch["scale"]).astype(F32)
        d_sum = (d_act * np.where(cch["sum_pre"] > 0, F32(1), F32(SLOPE))).astype(F32)
        d_bi = (d_act * np.where(cch["bi_pre"] > 0, F32(1), F32(SLOPE))).astype(F32)
        g[f"GC_weights.{l}.weight"] = (d_sum.T @ cch["side"]).astype(F32)
        g[f"GC_weights.{l}.bias"] = d_sum.sum(0, dtype=F32)
        g[f"Bi_weights.{l}.weight"] = (d_bi.T @ cch["bi_in"]).astype(F32)
        g[f"Bi_weights.{l}.bias"] = d_bi.sum(0, dtype=F32)
        if terms is not None:
            terms[f"GC_weights.{l}.bias"] = float(np.abs(d_sum).sum(0).max())
            terms[f"Bi_weights.{l}.bias"] = float(np.abs(d_bi).sum(0).max())
        d_side = (d_sum @ w[f"GC_weights.{l}.weight"]).astype(F32)
        d_bi_in = (d_bi @ w[f"Bi_weights.{l}.weight"]).astype(F32)
        d_ego = (d_bi_in * cch["side"]).astype(F32)
        d_side = (d_side + d_bi_in * cch["ego"]).astype(F32)
        d_next = (d_ego + adj_t @ d_side).astype(F32)
    d_e0 = d_all[:, :widths[0]] + (d_next if d_next is not None else 0)
    g["user_embedding.weight"] = d_e0[:U].astype(F32)
    g["item_embedding.weight"] = d_e0[U:].astype(F32)
    return float(loss), {k: g[k] for k in keys(L)}


def ngcf_train_step(w, st, adj, batch, decay, batch_size, optimizer="adam", lr=0.05, masks=None, drop=None):
    """NGCFEngine.train_single_batch: returns the loss (the reference also returns reg_loss = 0.0)."""
    loss, g = ngcf_grads(w, adj, batch[0], batch[1], batch[2], decay, batch_size, masks, drop)
    opt_step(w, g, st, optimizer, lr)
    return loss


def ngcf_predict(w, adj, users, items):
    """NGCF.predict (ngcf.py:82-100), eval mode (no dropout): dot product of the concatenated rows."""
    allv, _ = ngcf_forward(w, adj)
    U = w["user_embedding.weight"].shape[0]
    return (allv[np.asarray(users)] * allv[U + np.asarray(items)]).sum(1, dtype=F32)
