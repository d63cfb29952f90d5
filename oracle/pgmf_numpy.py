"""ORACLE (test infrastructure — never imported by the product path).

numpy fp32 restatement of the PairwiseGMF training step (the CMN pre-training model).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference lines (relative to /root/reference/beta_rec/):
    models/pairwise_gmf.py:10-17    truncated_normal_ (initialisation)
    models/pairwise_gmf.py:28-46    parameters user_memory / item_memory / v
    models/pairwise_gmf.py:48-62    forward: relu(v(u * i))
    models/pairwise_gmf.py:82-116   train_single_batch: loss + lambda ||v||, backward, clip, step
    models/pairwise_gmf.py:144-158  bpr_loss with eps inside the log
Pinned against golden vectors captured from the real reference by ``oracle/gen_golden.py --pgmf``
(tests/golden/pgmf_*.npz); see tests/test_oracle_golden_pgmf.py.

Parameters are a dict with the reference's state_dict keys:
    user_memory.weight [U,D], item_memory.weight [I,D], v.weight [1,D]
"""
import numpy as np

from .mf_numpy import F32, new_opt_state, opt_step, sigmoid  # noqa: F401  (shared optimizer arithmetic)

KEYS = ("user_memory.weight", "item_memory.weight", "v.weight")
EPS = F32(1e-12)


def pgmf_scores(w, users, items):
    """PairwiseGMF.forward for one item list: (relu(pre), pre)."""
    u = w["user_memory.weight"][users]
    i = w["item_memory.weight"][items]
    pre = ((u * i) @ w["v.weight"][0]).astype(F32)
    return np.maximum(pre, F32(0)), pre


def pgmf_grads(w, users, pos, neg, l2_lambda):
    """forward + loss + backward of train_single_batch, before the clip: (loss, grads)."""
    B = F32(len(users))
    v = w["v.weight"][0]
    u = w["user_memory.weight"][users]
    p = w["item_memory.weight"][pos]
    n = w["item_memory.weight"][neg]
    sp, pre_p = pgmf_scores(w, users, pos)
    sn, pre_n = pgmf_scores(w, users, neg)
    y = sigmoid(sp - sn)
    l2 = np.sqrt((v * v).sum(dtype=F32), dtype=F32)
    loss = (-np.log(y + EPS, dtype=F32)).mean(dtype=F32) + F32(l2_lambda) * l2
    # mean -> (-1 *) -> log -> (+ eps) -> sigmoid -> relu, each as its autograd backward
    dx = (-(F32(1) / B) / (y + EPS)) * ((F32(1) - y) * y)
    dp = np.where(pre_p > 0, dx, F32(0)).astype(F32)
    dn = np.where(pre_n > 0, -dx, F32(0)).astype(F32)
    g = {k: np.zeros_like(w[k]) for k in KEYS}
    np.add.at(g["user_memory.weight"], users, (dp[:, None] * p + dn[:, None] * n) * v)
    np.add.at(g["item_memory.weight"], pos, dp[:, None] * (u * v))
    np.add.at(g["item_memory.weight"], neg, dn[:, None] * (u * v))
    gv = (dp[:, None] * (u * p) + dn[:, None] * (u * n)).sum(axis=0, dtype=F32)
    g["v.weight"][0] = gv + (F32(l2_lambda) / (F32(2) * l2)) * (F32(2) * v)
    return float(loss), g


def clip_grad_norm(g, max_norm):
    """torch.nn.utils.clip_grad_norm_ (L2): scales ``g`` in place, returns the total norm."""
    norms = np.array([np.sqrt((g[k].astype(F32) ** 2).sum(dtype=F32), dtype=F32) for k in g], dtype=F32)
    total = np.sqrt((norms * norms).sum(dtype=F32), dtype=F32)
    coef = F32(max_norm) / (total + F32(1e-6))
    coef = F32(min(coef, F32(1.0)))
    for k in g:
        g[k] *= coef
    return float(total)


def pgmf_train_step(w, st, batch, optimizer="adam", lr=1e-4, l2_lambda=1e-4, grad_clip=5.0):
    """PairwiseGMFEngine.train_single_batch: returns the loss; ``w`` and ``st`` move in place."""
    loss, g = pgmf_grads(w, batch[0], batch[1], batch[2], l2_lambda)
    clip_grad_norm(g, grad_clip)
    opt_step(w, g, st, optimizer, lr)
    return loss


def truncated_normal(shape, rng_normal, mean=0.0, std=1.0):
    """models/pairwise_gmf.py:10-17 given a callable that draws N(0,1) of a shape: of 4 candidates per
    element keep the first inside (-2, 2) (the first one when none is)."""
    tmp = rng_normal(tuple(shape) + (4,))
    valid = (tmp < 2) & (tmp > -2)
    ind = np.argmax(valid, axis=-1)[..., None]
    return (np.take_along_axis(tmp, ind, axis=-1)[..., 0] * F32(std) + F32(mean)).astype(F32)
