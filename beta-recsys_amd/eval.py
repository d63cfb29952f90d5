"""On-device evaluation: the caller that follows the training step.

Mirrors the two functions of beta_rec/core/eval_engine.py the training loop uses between epochs:

* ``predict`` — ``EvalEngine.predict`` (eval_engine.py:231-274): score every (user, item) row of a
  frame with ``model.predict`` (the HIP scoring kernels).
* ``evaluate`` — module-level ``evaluate`` (eval_engine.py:49-87): ranking metrics
  ``ndcg / map / precision / recall @ k`` of ``utils/evaluation.py:535-752``, here computed by
  ``hiprec_rank_metrics`` (csrc/eval.hip) without the pandas groupby/merge round trip.

There is no CPU fallback: the arithmetic runs in libhiprec.so on the GPU or not at all.  torch is
used for memory and for grouping the rows by user (a stable sort — plumbing).
"""
import ctypes

import numpy as np
import torch

from . import _lib

DEFAULT_USER_COL = "col_user"          # beta_rec/utils/constants.py:1-3
DEFAULT_ITEM_COL = "col_item"
DEFAULT_RATING_COL = "col_rating"

RANK_METRICS = ("precision", "recall", "ndcg", "map")   # column order of hiprec_rank_metrics
MAX_K = 8                                               # HIPREC_RANK_MAX_K


def _column(frame, name):
    """A pandas DataFrame column, a dict entry, or an attribute -> numpy / tensor."""
    col = frame[name]
    return col.to_numpy() if hasattr(col, "to_numpy") else col


def _resolve_device(device):
    dev = torch.device("cuda" if device is None else device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError(f"hiprec evaluation runs on an MI355X through libhiprec.so; device {dev} has "
                           "no HIP path and there is deliberately no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device() if dev.index is None else dev.index)


def group_by_user(users):
    """(order | None, seg_ptr): rows grouped by user keeping their relative order.

    ``order`` is None when the frame is already grouped (the usual layout of the reference's test
    frames); ``seg_ptr`` has one entry per user plus one."""
    n = users.numel()
    if n == 0:
        return None, torch.zeros(1, dtype=torch.int64, device=users.device)
    order = None
    if n > 1 and bool((users[1:] < users[:-1]).any()):
        users, order = torch.sort(users, stable=True)
    _, counts = torch.unique_consecutive(users, return_counts=True)
    seg_ptr = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=users.device)
    torch.cumsum(counts, 0, out=seg_ptr[1:])
    return order, seg_ptr


def rank_metrics(users, ratings, scores, k_list, device=None):
    """(n_common_users, table[len(k_list), 4] float64 numpy; columns precision, recall, ndcg, map).

    Inputs are the aligned columns of the frame ``evaluate`` receives (any of numpy / list / tensor);
    scores are ranked in fp32, the dtype every engine's ``predict`` returns."""
    k_list = [int(k) for k in k_list]
    if not 1 <= len(k_list) <= MAX_K:
        raise ValueError(f"between 1 and {MAX_K} cut-offs per call, got {len(k_list)}")
    if min(k_list) < 1:
        raise ValueError(f"k must be >= 1, got {k_list}")
    if torch.is_tensor(scores) and scores.device.type == "cuda" and device is None:
        device = scores.device
    dev = _resolve_device(device)
    lib = _lib.load()
    users_t = torch.as_tensor(np.asarray(users) if not torch.is_tensor(users) else users).to(dev, torch.int64).reshape(-1)
    ratings_t = torch.as_tensor(np.asarray(ratings) if not torch.is_tensor(ratings) else ratings).to(dev, torch.float32).reshape(-1)
    scores_t = torch.as_tensor(np.asarray(scores) if not torch.is_tensor(scores) else scores).to(dev, torch.float32).reshape(-1)
    if not (users_t.numel() == ratings_t.numel() == scores_t.numel()):
        raise ValueError("users, ratings and predictions must have one entry per row of the frame "
                         f"({users_t.numel()}, {ratings_t.numel()}, {scores_t.numel()})")
    order, seg_ptr = group_by_user(users_t)
    if order is not None:
        ratings_t, scores_t = ratings_t[order], scores_t[order]
    ratings_t, scores_t = ratings_t.contiguous(), scores_t.contiguous()
    n_seg = seg_ptr.numel() - 1
    n_k = len(k_list)
    ws_bytes = lib.hiprec_rank_metrics_workspace_bytes(n_seg, n_k)
    workspace = torch.empty(max(ws_bytes // 8, 1), dtype=torch.float64, device=dev)
    out = torch.empty(1 + 4 * n_k, dtype=torch.float64, device=dev)
    ks = (ctypes.c_int32 * n_k)(*k_list)
    with torch.cuda.device(dev):
        _lib.check(lib.hiprec_rank_metrics(_lib.ptr(seg_ptr), n_seg, _lib.ptr(scores_t), _lib.ptr(ratings_t),
                                           ks, n_k, _lib.ptr(workspace), ws_bytes, _lib.ptr(out),
                                           _lib.stream_ptr(dev)))
    host = out.cpu().numpy()
    return int(host[0]), host[1:].reshape(n_k, 4).copy()


def evaluate(data_df, predictions, metrics, k_li, device=None):
    """eval_engine.py:49-87: ``{f"{metric}@{k}": value}`` for every k in ``k_li`` and metric name.

    ``data_df`` is a pandas DataFrame (or any mapping) with ``col_user`` and ``col_rating`` columns,
    ``predictions`` the scores of its rows (numpy, or a device tensor straight from ``model.predict``).
    Only the ranking metrics of the training loop are offered; rating metrics (rmse, mae, ...) raise."""
    unknown = [m for m in metrics if m not in RANK_METRICS]
    if unknown:
        raise KeyError(f"metrics {unknown} are not ranking metrics; supported: {RANK_METRICS}")
    if type(k_li) != list:  # noqa: E721  (the reference's own test, eval_engine.py:81-82)
        k_li = [k_li]
    users = _column(data_df, DEFAULT_USER_COL)
    ratings = _column(data_df, DEFAULT_RATING_COL)
    result = {}
    for lo in range(0, len(k_li), MAX_K):
        chunk = k_li[lo:lo + MAX_K]
        _, table = rank_metrics(users, ratings, predictions, chunk, device=device)
        for j, k in enumerate(chunk):
            for metric in metrics:
                result[f"{metric}@{k}"] = float(table[j, RANK_METRICS.index(metric)])
    # key order of the reference: k outer, metric inner
    return {f"{m}@{k}": result[f"{m}@{k}"] for k in k_li for m in metrics}


def predict_device(data_df, model, batch_size=None):
    """Scores of every row as ONE device tensor (no host copy), optionally in batches."""
    user_ids = np.asarray(_column(data_df, DEFAULT_USER_COL))
    item_ids = np.asarray(_column(data_df, DEFAULT_ITEM_COL))
    n = len(user_ids)
    if not batch_size or batch_size >= n:
        return model.predict(user_ids, item_ids).flatten().detach()
    parts = [model.predict(user_ids[s:s + batch_size], item_ids[s:s + batch_size]).flatten().detach()
             for s in range(0, n, batch_size)]
    return torch.cat(parts)


def predict(data_df, model, batch_eval=False, batch_size=None):
    """EvalEngine.predict (eval_engine.py:231-274): numpy scores of the frame's rows.

    With ``batch_eval`` the frame is scored ``batch_size`` rows at a time; the reference's batch
    loop has an off-by-one that drops no row (its ``stop_batch`` branch, :251-253), so plain
    slicing yields the same array."""
    if batch_eval and not batch_size:
        raise ValueError("batch_eval=True needs batch_size (EvalEngine takes config['model']['batch_size'])")
    scores = predict_device(data_df, model, batch_size if batch_eval else None)
    out = scores.to(torch.device("cpu")).numpy()
    # the reference's batched branch builds the array with np.append on an empty float64 array
    return out.astype(np.float64) if batch_eval else out
