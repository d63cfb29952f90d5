"""ORACLE (test infrastructure — never imported by the product path).

numpy fp32 restatement of the beta-recsys MF training step.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.

Each function cites the reference lines (relative to /root/reference/beta_rec/) it restates.  The
restatement is pinned against golden vectors captured from the real reference by
oracle/gen_golden.py (tests/golden/mf_*.npz); see tests/test_oracle_golden.py.

Parameters are a dict with the reference's state_dict keys:
    global_bias [1], user_emb.weight [U,D], item_emb.weight [I,D], user_bias.weight [U,1],
    item_bias.weight [I,1]            (models/mf.py:21-25)
"""
import numpy as np

F32 = np.float32
KEYS = ("global_bias", "user_emb.weight", "item_emb.weight", "user_bias.weight", "item_bias.weight")


def sigmoid(x):
    """torch.sigmoid in fp32."""
    x = x.astype(F32)
    return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


def logsigmoid(x):
    """F.logsigmoid as ATen computes it: min(x,0) - log1p(exp(-|x|))."""
    x = x.astype(F32)
    return (np.minimum(x, F32(0)) - np.log1p(np.exp(-np.abs(x), dtype=F32), dtype=F32)).astype(F32)


def mf_forward(w, users, items):
    """MF.forward, models/mf.py:32-55 -> (scores [B], regularizer scalar, cache)."""
    u = w["user_emb.weight"][users]
    i = w["item_emb.weight"][items]
    bu = w["user_bias.weight"][users][:, 0]
    bi = w["item_bias.weight"][items][:, 0]
    s = (u * i).sum(axis=1, dtype=F32) + bu + bi + w["global_bias"][0]
    y = sigmoid(s)
    B = F32(len(users))
    reg = ((u ** 2).sum(dtype=F32) + (i ** 2).sum(dtype=F32) + (bu ** 2).sum(dtype=F32)
           + (bi ** 2).sum(dtype=F32)) / B
    return y, F32(reg), (u, i, bu, bi)


def zeros_like_params(w):
    return {k: np.zeros_like(v) for k, v in w.items()}


def _scatter_forward_grads(g, users, items, ds, cache, reg_coef, B):
    """Backward of one MF.forward call given d(loss)/d(pre-sigmoid score) = ds [B]."""
    u, i, bu, bi = cache
    rr = F32(2.0) * F32(reg_coef) / F32(B)
    np.add.at(g["user_emb.weight"], users, ds[:, None] * i + rr * u)
    np.add.at(g["item_emb.weight"], items, ds[:, None] * u + rr * i)
    np.add.at(g["user_bias.weight"][:, 0], users, ds + rr * bu)
    np.add.at(g["item_bias.weight"][:, 0], items, ds + rr * bi)
    g["global_bias"][0] += ds.sum(dtype=F32)


def mf_bpr_grads(w, users, pos, neg, reg_coef=0.0, global_batch=None):
    """BPR branch of MFEngine.train_single_batch (models/mf.py:101-107,116-117) + bpr_loss
    (models/torch_engine.py:104-105): returns (loss, regularizer, dense grads).

    ``global_batch``: when these triples are only one rank's part of a larger batch, the mean is
    over that global batch (the returned loss / regularizer / grads are then this part's share)."""
    users, pos, neg = (np.asarray(a, dtype=np.int64) for a in (users, pos, neg))
    B = len(users) if global_batch is None else int(global_batch)
    yp, reg_p, cp = mf_forward(w, users, pos)
    yn, reg_n, cn = mf_forward(w, users, neg)
    if global_batch is not None:
        scale = F32(len(users)) / F32(B)
        reg_p, reg_n = F32(reg_p * scale), F32(reg_n * scale)
    x = (yp - yn).astype(F32)
    loss = F32(-logsigmoid(x).sum(dtype=F32) / F32(B))
    # d/dx of -mean(logsigmoid(x)) = -sigmoid(-x)/B
    delta = (-sigmoid(-x) / F32(B)).astype(F32)
    d_pos = (delta * (yp * (F32(1) - yp))).astype(F32)
    d_neg = (-delta * (yn * (F32(1) - yn))).astype(F32)
    g = zeros_like_params(w)
    _scatter_forward_grads(g, users, pos, d_pos, cp, reg_coef, B)
    _scatter_forward_grads(g, users, neg, d_neg, cn, reg_coef, B)
    return loss, F32(reg_p + reg_n), g


def mf_bce_grads(w, users, items, ratings, reg_coef=0.0):
    """BCE branch (models/mf.py:108-111) + bce_loss (models/torch_engine.py:108-121):
    torch.nn.BCELoss(mean) with its log clamp at -100 and ATen's backward epsilon 1e-12."""
    users, items = (np.asarray(a, dtype=np.int64) for a in (users, items))
    r = np.asarray(ratings, dtype=F32)
    B = len(users)
    y, reg, cache = mf_forward(w, users, items)
    ly = np.maximum(np.log(y, dtype=F32), F32(-100))
    l1y = np.maximum(np.log1p(-y, dtype=F32), F32(-100))
    loss = F32((-(r * ly + (F32(1) - r) * l1y)).mean(dtype=F32))
    gy = ((y - r) / np.maximum((F32(1) - y) * y, F32(1e-12)) / F32(B)).astype(F32)
    ds = (gy * ((F32(1) - y) * y)).astype(F32)
    g = zeros_like_params(w)
    _scatter_forward_grads(g, users, items, ds, cache, reg_coef, B)
    return loss, F32(reg), g


# ---- optimizers (models/torch_engine.py:23-39: torch defaults, only lr set) ----------------------

def new_opt_state(w, optimizer):
    st = {"step": 0}
    if optimizer == "adam":
        st["exp_avg"] = zeros_like_params(w)
        st["exp_avg_sq"] = zeros_like_params(w)
    elif optimizer == "rmsprop":
        st["square_avg"] = zeros_like_params(w)
    elif optimizer != "sgd":
        raise ValueError(optimizer)
    return st


def opt_step(w, g, st, optimizer, lr):
    """One dense optimizer step over every parameter tensor, in place."""
    st["step"] += 1
    t = st["step"]
    if optimizer == "sgd":  # torch.optim.SGD, momentum 0: p.add_(g, alpha=-lr)
        for k in w:
            w[k] -= F32(lr) * g[k]
    elif optimizer == "adam":  # torch/optim/adam.py::_single_tensor_adam, betas (0.9,0.999) eps 1e-8
        b1, b2, eps = 0.9, 0.999, 1e-8
        bc1 = 1.0 - b1 ** t
        bc2 = 1.0 - b2 ** t
        step_size = F32(lr / bc1)
        bc2_sqrt = F32(bc2 ** 0.5)
        for k in w:
            m, v = st["exp_avg"][k], st["exp_avg_sq"][k]
            m += F32(1.0 - b1) * (g[k] - m)
            v *= F32(b2)
            v += F32(1.0 - b2) * g[k] * g[k]
            denom = np.sqrt(v, dtype=F32) / bc2_sqrt + F32(eps)
            w[k] += (-step_size * m) / denom
    elif optimizer == "rmsprop":  # torch/optim/rmsprop.py, alpha 0.99 eps 1e-8
        alpha, eps = 0.99, 1e-8
        for k in w:
            v = st["square_avg"][k]
            v *= F32(alpha)
            v += F32(1.0 - alpha) * g[k] * g[k]
            avg = np.sqrt(v, dtype=F32) + F32(eps)
            w[k] += (F32(-lr) * g[k]) / avg
    else:
        raise ValueError(optimizer)


def mf_train_step(w, st, batch, loss="bpr", optimizer="sgd", lr=0.05, reg_coef=0.0):
    """MFEngine.train_single_batch (models/mf.py:92-119): returns (loss, regularizer)."""
    if loss == "bpr":
        lval, reg, g = mf_bpr_grads(w, batch[0], batch[1], batch[2], reg_coef)
    elif loss == "bce":
        lval, reg, g = mf_bce_grads(w, batch[0], batch[1], batch[2], reg_coef)
    else:
        raise RuntimeError(f"Unsupported loss type {loss}")
    opt_step(w, g, st, optimizer, lr)
    return float(lval), float(reg)


def mf_predict(w, users, items):
    """MF.predict (models/mf.py:57-70)."""
    y, _, _ = mf_forward(w, np.asarray(users, dtype=np.int64), np.asarray(items, dtype=np.int64))
    return y


def init_params(n_users, n_items, dim, seed=0, stddev=0.1):
    """Random parameters with MF.__init__'s distribution (models/mf.py:26-30); biases start at 0
    but are perturbed a little here so that tests exercise them."""
    rng = np.random.default_rng(seed)
    return {
        "global_bias": np.array([0.05], dtype=F32),
        "user_emb.weight": (rng.standard_normal((n_users, dim)) * stddev).astype(F32),
        "item_emb.weight": (rng.standard_normal((n_items, dim)) * stddev).astype(F32),
        "user_bias.weight": (rng.standard_normal((n_users, 1)) * 0.01).astype(F32),
        "item_bias.weight": (rng.standard_normal((n_items, 1)) * 0.01).astype(F32),
    }


def copy_params(w):
    return {k: v.copy() for k, v in w.items()}
