"""ORACLE (test infrastructure — never imported by the product path).

numpy fp32 restatement of the beta-recsys NCF family training step (NeuMF, GMF, MLP); tower dropout
through explicit keep masks (``masks``: one boolean [B, width] array per Linear's input, kept entries
scaled by 1 / (1 - p); None = dropout 0 or eval mode).  Each function cites the reference lines (relative to /root/reference/beta_rec/).
Pinned against golden vectors captured from the real reference (oracle/gen_golden.py ->
tests/golden/ncf_*.npz, gmf_*.npz, mlp_*.npz; checked by tests/test_oracle_golden_ncf.py).

Parameters are dicts keyed like the reference's state_dict:
  NeuMF (models/ncf.py:24-50): embedding_{user,item}_{mlp,mf}.weight, fc_layers.{1,4,7,...}.{weight,
        bias}, affine_output.{weight,bias}
  GMF   (models/gmf.py:19-27): embedding_{user,item}.weight, affine_output.{weight,bias}
  MLP   (models/mlp.py:22-38): embedding_{user,item}.weight, fc_layers.*, affine_output.*
"""
import numpy as np

from .mf_numpy import F32, opt_step, sigmoid  # noqa: F401  (same optimizers / sigmoid)


def layer_keys(w):
    """Indices of the nn.Linear entries inside fc_layers (1, 4, 7, ...), in order."""
    idx = sorted({int(k.split(".")[1]) for k in w if k.startswith("fc_layers.") and k.endswith(".weight")})
    return idx


def bce(y, r):
    """torch.nn.BCELoss(mean) forward and d loss / d y as ATen computes them."""
    y, r = y.astype(F32), r.astype(F32)
    B = F32(len(y))
    ly = np.maximum(np.log(y, dtype=F32), F32(-100))
    l1y = np.maximum(np.log1p(-y, dtype=F32), F32(-100))
    loss = F32((-(r * ly + (F32(1) - r) * l1y)).mean(dtype=F32))
    gy = ((y - r) / np.maximum((F32(1) - y) * y, F32(1e-12)) / B).astype(F32)
    return loss, gy


def tower_forward(w, x, relu_input, masks=None, dropout=0.0):
    """The fc_layers loop.  NeuMF (models/ncf.py:64-66) applies an extra ReLU after EVERY entry of
    fc_layers, the Dropout entries included (quirk Q7), so the raw concatenated embedding is
    ReLU-ed before the first Linear (relu_input=True; ReLU and the non-negative dropout scaling
    commute).  MLP (models/mlp.py:47-48) applies the entries only.  Every Linear's input goes through
    its Dropout entry first (ncf.py:42-45): ``acts`` holds what the Linear actually saw."""
    acts = []
    h = np.maximum(x, F32(0)) if relu_input else x
    for l, li in enumerate(layer_keys(w)):
        if masks is not None and masks[l] is not None:
            h = (h * (masks[l].astype(F32) / F32(1.0 - dropout))).astype(F32)
        acts.append(h)
        z = (h @ w[f"fc_layers.{li}.weight"].T + w[f"fc_layers.{li}.bias"]).astype(F32)
        h = np.maximum(z, F32(0))
    return h, acts


def tower_backward(w, g, acts, h_out, dh, x, relu_input, masks=None, dropout=0.0):
    """Backward through the tower; returns d loss / d x (the concatenated embedding)."""
    keys = layer_keys(w)
    out = h_out
    for l in range(len(keys) - 1, -1, -1):
        li, h_in = keys[l], acts[l]
        dz = (dh * (out > 0)).astype(F32)
        g[f"fc_layers.{li}.weight"] += (dz.T @ h_in).astype(F32)
        g[f"fc_layers.{li}.bias"] += dz.sum(axis=0, dtype=F32)
        dh = (dz @ w[f"fc_layers.{li}.weight"]).astype(F32)
        if masks is not None and masks[l] is not None:
            dh = (dh * (masks[l].astype(F32) / F32(1.0 - dropout))).astype(F32)
        # the previous layer's ReLU output is h_in before its dropout; a dropped entry has dh = 0 already,
        # a kept one is > 0 exactly when the undropped activation was
        out = h_in
    if relu_input:
        dh = (dh * (x > 0)).astype(F32)
    return dh


def ncf_grads(w, users, items, ratings, kind="neumf", masks=None, dropout=0.0):
    """zero_grad + forward + BCELoss + backward of {NeuMF,GMF,MLP}Engine.train_single_batch
    (models/ncf.py:100-120, models/gmf.py:60-80, models/mlp.py:76-96).  Returns (loss, grads)."""
    users = np.asarray(users, dtype=np.int64)
    items = np.asarray(items, dtype=np.int64)
    r = np.asarray(ratings, dtype=F32)
    g = {k: np.zeros_like(v) for k, v in w.items()}
    if kind == "neumf":
        um, im = w["embedding_user_mlp.weight"][users], w["embedding_item_mlp.weight"][items]
        ug, ig = w["embedding_user_mf.weight"][users], w["embedding_item_mf.weight"][items]
        x = np.concatenate([um, im], axis=1)                       # ncf.py:59-61
        h, acts = tower_forward(w, x, True, masks, dropout)         # ncf.py:64-66
        mf = (ug * ig).astype(F32)                                  # ncf.py:62
        vec = np.concatenate([h, mf], axis=1)                       # ncf.py:68
    elif kind == "gmf":
        ug, ig = w["embedding_user.weight"][users], w["embedding_item.weight"][items]
        vec = (ug * ig).astype(F32)                                 # gmf.py:33
    elif kind == "mlp":
        um, im = w["embedding_user.weight"][users], w["embedding_item.weight"][items]
        x = np.concatenate([um, im], axis=1)                        # mlp.py:44-46
        h, acts = tower_forward(w, x, False, masks, dropout)        # mlp.py:47-48
        vec = h
    else:
        raise ValueError(kind)
    wo, bo = w["affine_output.weight"], w["affine_output.bias"]
    logits = (vec @ wo.T + bo).astype(F32)[:, 0]
    y = sigmoid(logits)
    loss, gy = bce(y, r)
    dlogit = (gy * ((F32(1) - y) * y)).astype(F32)                 # sigmoid_backward
    g["affine_output.weight"] += (dlogit[None, :] @ vec).astype(F32)
    g["affine_output.bias"] += dlogit.sum(dtype=F32)
    dvec = (dlogit[:, None] * wo).astype(F32)
    if kind == "neumf":
        E = ug.shape[1]
        dh, dmf = dvec[:, :-E], dvec[:, -E:]
    elif kind == "gmf":
        dh, dmf = None, dvec
    else:
        dh, dmf = dvec, None
    if dmf is not None:
        ku = "embedding_user_mf.weight" if kind == "neumf" else "embedding_user.weight"
        ki = "embedding_item_mf.weight" if kind == "neumf" else "embedding_item.weight"
        np.add.at(g[ku], users, dmf * ig)
        np.add.at(g[ki], items, dmf * ug)
    if dh is not None:
        dx = tower_backward(w, g, acts, h, dh, x, kind == "neumf", masks, dropout)
        Dm = um.shape[1]
        ku = "embedding_user_mlp.weight" if kind == "neumf" else "embedding_user.weight"
        ki = "embedding_item_mlp.weight" if kind == "neumf" else "embedding_item.weight"
        np.add.at(g[ku], users, dx[:, :Dm])
        np.add.at(g[ki], items, dx[:, Dm:])
    return float(loss), g, y


def ncf_predict(w, users, items, kind="neumf"):
    """model.forward under no_grad (ncf.py:73-78): [n, 1] sigmoid scores."""
    _, _, y = ncf_grads(w, users, items, np.zeros(len(users), dtype=F32), kind)
    return y[:, None]


def new_opt_state(w, optimizer):
    st = {"step": 0}
    z = lambda: {k: np.zeros_like(v) for k, v in w.items()}  # noqa: E731
    if optimizer == "adam":
        st["exp_avg"], st["exp_avg_sq"] = z(), z()
    elif optimizer == "rmsprop":
        st["square_avg"] = z()
    return st


def ncf_train_step(w, st, batch, kind="neumf", optimizer="adam", lr=1e-3):
    loss, g, _ = ncf_grads(w, batch[0], batch[1], batch[2], kind)
    opt_step(w, g, st, optimizer, lr)
    return loss
