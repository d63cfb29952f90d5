"""ORACLE (test infrastructure — never imported by the product path).

numpy fp32 restatement of the Triple2vec training step, its predict and (bit for bit) the device
alias sampler.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Reference lines (relative to /root/reference/beta_rec/):
    models/triple2vec.py:11-34    parameters and init (uniform(-0.01, 0.01) tables, zero biases)
    models/triple2vec.py:36-92    forward (= the batch loss)
    models/triple2vec.py:94-104   predict
    models/triple2vec.py:115-124  train_single_batch
    utils/alias_table.py:11-97    AliasTable construction and sample
Pinned against golden vectors captured from the real reference by ``oracle/gen_golden.py --t2v``
(tests/golden/t2v_*.npz); see tests/test_oracle_golden_t2v.py.

Parameters are a dict with the reference's state_dict keys:
    user_emb.weight [U,D], item_emb1.weight [I,D], item_emb2.weight [I,D], user_bias.weight [U,1],
    item_bias.weight [I,1]
``shared`` (default True) restates triple2vec.py:19,38-39: ``use_bias = n_neg`` is truthy for every
usable config, so ``item_emb2`` IS ``item_emb1`` from the first forward on; item_emb2.weight then
mirrors item_emb1.weight and its own gradient slot stays zero.
"""
import numpy as np

from .mf_numpy import F32, logsigmoid, new_opt_state, opt_step, sigmoid  # noqa: F401
from .sampler_numpy import splitmix64

KEYS = ("user_emb.weight", "item_emb1.weight", "item_emb2.weight", "user_bias.weight", "item_bias.weight")


def t2v_grads(w, batch, batch_size, shared=True, acc=F32):
    """Loss of Triple2vec.forward and its gradient: (loss, grads).  ``batch`` = (pos_u [B], pos_i_1 [B],
    pos_i_2 [B], neg_u [B,n], neg_i_1 [B,n], neg_i_2 [B,n]); ``batch_size`` is the CONFIGURED one
    (triple2vec.py:92 divides by 3 * self.batch_size whatever the batch's real length).
    ``acc``: dtype the scattered terms are summed in.  fp32 (default) sums in batch order like the
    reference's index_add; fp64 gives the order-independent sum, used at sizes where one hot row
    receives thousands of terms and the ORDER noise of fp32 would otherwise be part of the yardstick."""
    pu, p1, p2, nu, n1, n2 = (np.asarray(x, dtype=np.int64) for x in batch)
    U, E1 = w["user_emb.weight"], w["item_emb1.weight"]
    E2 = E1 if shared else w["item_emb2.weight"]
    bu, bi = w["user_bias.weight"][:, 0], w["item_bias.weight"][:, 0]
    c = F32(1.0 / (3 * batch_size))
    eu, e1, e2 = U[pu], E1[p1], E2[p2]
    ru, r1, r2 = U[nu], E1[n2], E2[n2]        # triple2vec.py:45-47: both item rows by neg_i_2
    x = (eu * (e1 + e2)).sum(1, dtype=F32) + bu[pu]
    x1 = (e1 * (eu + e2)).sum(1, dtype=F32) + bi[p1]
    x2 = (e2 * (eu + e1)).sum(1, dtype=F32) + bi[p2]
    y = np.einsum("bnd,bd->bn", ru, eu).astype(F32) + bu[nu]
    y1 = np.einsum("bnd,bd->bn", r1, e1).astype(F32) + bi[n1]   # bias by neg_i_1 (triple2vec.py:70)
    y2 = np.einsum("bnd,bd->bn", r2, e2).astype(F32) + bi[n2]
    total = F32(0)
    for pos_s, neg_s in ((x, y), (x1, y1), (x2, y2)):
        total += -(logsigmoid(pos_s).sum(dtype=F32) + logsigmoid(-neg_s).sum(dtype=F32))
    loss = total / F32(3 * batch_size)
    dx, dx1, dx2 = (-sigmoid(-v) * c for v in (x, x1, x2))
    dy, dy1, dy2 = (sigmoid(v) * c for v in (y, y1, y2))
    g = {k: np.zeros(w[k].shape, dtype=acc) for k in KEYS}
    gU, gE1 = g["user_emb.weight"], g["item_emb1.weight"]
    gE2 = gE1 if shared else g["item_emb2.weight"]
    gbu, gbi = g["user_bias.weight"][:, 0], g["item_bias.weight"][:, 0]
    np.add.at(gU, pu, dx[:, None] * (e1 + e2) + dx1[:, None] * e1 + dx2[:, None] * e2
              + np.einsum("bn,bnd->bd", dy, ru).astype(F32))
    np.add.at(gE1, p1, dx[:, None] * eu + dx1[:, None] * (eu + e2) + dx2[:, None] * e2
              + np.einsum("bn,bnd->bd", dy1, r1).astype(F32))
    np.add.at(gE2, p2, dx[:, None] * eu + dx1[:, None] * e1 + dx2[:, None] * (eu + e1)
              + np.einsum("bn,bnd->bd", dy2, r2).astype(F32))
    np.add.at(gU, nu, dy[:, :, None] * eu[:, None, :])
    np.add.at(gE1, n2, dy1[:, :, None] * e1[:, None, :])
    np.add.at(gE2, n2, dy2[:, :, None] * e2[:, None, :])
    np.add.at(gbu, pu, dx)
    np.add.at(gbi, p1, dx1)
    np.add.at(gbi, p2, dx2)
    np.add.at(gbu, nu, dy)
    np.add.at(gbi, n1, dy1)
    np.add.at(gbi, n2, dy2)
    return float(loss), {k: v.astype(F32) for k, v in g.items()}


def t2v_train_step(w, st, batch, batch_size, optimizer="adam", lr=5e-4, shared=True, acc=F32):
    """Triple2vecEngine.train_single_batch: returns the loss; ``w`` and ``st`` move in place.  With
    ``shared`` the orphaned item_emb2 parameter gets no gradient (torch skips it) and the visible
    item_emb2.weight follows item_emb1.weight."""
    loss, g = t2v_grads(w, batch, batch_size, shared, acc)
    opt_step(w, g, st, optimizer, lr)
    if shared:
        w["item_emb2.weight"][...] = w["item_emb1.weight"]
    return loss


def t2v_predict(w, users, items, shared=True):
    """Triple2vec.predict, triple2vec.py:94-104."""
    E2 = w["item_emb1.weight"] if shared else w["item_emb2.weight"]
    mid = (w["item_emb1.weight"][items] + E2[items]) / F32(2)
    return (w["user_emb.weight"][users] * mid).sum(1, dtype=F32)


# ---- alias sampling --------------------------------------------------------------------------------
def build_alias_table(freq):
    """AliasTable.__init__ for a list of frequencies (alias_table.py:24-80): (prob_arr fp64, alias_arr)."""
    freq = np.asarray(freq, dtype=np.float64)
    n = len(freq)
    prob = n * (freq / freq.sum())
    alias = np.zeros(n, dtype=np.int64)
    small = [i for i in range(n) if prob[i] < 1.0]
    large = [i for i in range(n) if not prob[i] < 1.0]
    while small and large:
        s, l = small.pop(), large.pop()
        alias[s] = l
        prob[l] = prob[l] - (1.0 - prob[s])
        (small if prob[l] < 1.0 else large).append(l)
    return prob, alias


def alias_sample(prob, alias, labels, seed, n):
    """hiprec_alias_sample restated on uint64 arrays (vocab < 2^32 so that mulhi64 splits in halves)."""
    U64 = np.uint64
    vocab = len(prob)
    assert 0 < vocab < (1 << 32)
    with np.errstate(over="ignore"):
        h1 = splitmix64(U64(seed) ^ splitmix64(np.arange(n, dtype=U64)))
        h2 = splitmix64(h1)
        hi, lo = h1 >> U64(32), h1 & U64(0xFFFFFFFF)
        col = ((hi * U64(vocab) + ((lo * U64(vocab)) >> U64(32))) >> U64(32)).astype(np.int64)
    u = (h2 >> U64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    pick = np.where(u < np.asarray(prob)[col], col, np.asarray(alias, dtype=np.int64)[col])
    return pick if labels is None else np.asarray(labels, dtype=np.int64)[pick]
