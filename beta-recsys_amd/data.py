"""Device-side training-set construction: the data step in front of the hot path.

Mirrors the three loader builders of ``beta_rec/data/base_data.py`` that feed the engines of this
package — same inputs (the split's ``train`` frame, ``n_users``, ``n_items``), same row order and
tensor layout of the resulting dataset, batches of the same tuple shape:

* ``instance_bpr_loader``      (base_data.py:218-253)  (user, pos, neg) triples, one negative per row
* ``instance_bce_loader``      (base_data.py:182-216)  (user, item, rating): each row followed by its
                                                        ``num_negative`` negatives with rating 0
* ``instance_mul_neg_loader``  (base_data.py:254-288)  (user, pos, neg[num_negative])

The reference samples with ``random.sample`` inside a pandas ``iterrows`` loop on the host and then
lets a ``DataLoader`` index the tensors element by element; here the negatives come from
``hiprec_sample_negatives`` (csrc/sampler.hip: k distinct items, uniform over the items the user
never touched — the same distribution, a different random stream) and the batches from device-side
batchers.  No CPU fallback: the sampler runs in libhiprec.so or not at all.
"""
import numpy as np
import torch

from . import _lib
from .eval import DEFAULT_ITEM_COL, DEFAULT_RATING_COL, DEFAULT_USER_COL, _column, _resolve_device
from .mf import DeviceTripleBatcher, _new_stats, raise_on_status, read_stats


def _draw_seed():
    """A 62-bit seed from torch's global CPU generator, so ``torch.manual_seed`` controls the draw."""
    return int(torch.randint(0, 2**62, (1,)).item())


def build_positive_csr(users, items, n_users, n_items):
    """(user_ptr[n_users+1], pos_sorted): each user's positive items, ascending and unique
    (``groupby(user)[item].apply(set)``, base_data.py:227-231) — torch ops on the tensors' device."""
    users = users.to(torch.int64)
    items = items.to(torch.int64)
    if users.numel() and (int(users.min()) < 0 or int(users.max()) >= n_users
                          or int(items.min()) < 0 or int(items.max()) >= n_items):
        raise IndexError("training frame holds a user / item id outside [0, n_users) x [0, n_items)")
    key = torch.unique(users * n_items + items)          # sorted
    owner = torch.div(key, n_items, rounding_mode="floor")
    ptr = torch.zeros(n_users + 1, dtype=torch.int64, device=users.device)
    torch.cumsum(torch.bincount(owner, minlength=n_users), 0, out=ptr[1:])
    return ptr, (key - owner * n_items).contiguous()


def sample_negatives(users, items, n_users, n_items, k=1, seed=None, device=None):
    """int64 tensor [n_rows, k]: for every (user, item) row, k distinct items the user never interacted
    with anywhere in the frame, uniformly at random (a pure function of ``seed``)."""
    if k < 1:
        raise ValueError(f"k must be >= 1, got {k}")
    if torch.is_tensor(users) and users.device.type == "cuda" and device is None:
        device = users.device
    dev = _resolve_device(device)
    lib = _lib.load()
    users_t = torch.as_tensor(np.asarray(users) if not torch.is_tensor(users) else users).to(dev, torch.int64).reshape(-1).contiguous()
    items_t = torch.as_tensor(np.asarray(items) if not torch.is_tensor(items) else items).to(dev, torch.int64).reshape(-1).contiguous()
    if users_t.numel() != items_t.numel():
        raise ValueError("users and items differ in length")
    ptr, cols = build_positive_csr(users_t, items_t, int(n_users), int(n_items))
    out = torch.empty((users_t.numel(), int(k)), dtype=torch.int64, device=dev)
    stats = _new_stats(dev)
    seed = _draw_seed() if seed is None else int(seed)
    with torch.cuda.device(dev):
        _lib.check(lib.hiprec_sample_negatives(
            _lib.ptr(ptr), _lib.ptr(cols), int(n_users), int(n_items), _lib.ptr(users_t), users_t.numel(),
            int(k), seed, _lib.ptr(out), _lib.ptr(stats), _lib.stream_ptr(dev)))
    raise_on_status(read_stats(stats).status)
    return out


class DeviceTensorBatcher:
    """``DataLoader(TensorDataset-like, batch_size, shuffle=True)`` over device-resident tensors:
    iterating yields tuples of batch slices, a fresh device-side permutation per epoch."""

    def __init__(self, tensors, batch_size, shuffle=True):
        self.tensors = tuple(tensors)
        if len({t.shape[0] for t in self.tensors}) != 1:
            raise ValueError("tensors differ in length")
        self.batch_size = int(batch_size)
        self.shuffle = shuffle

    def __len__(self):
        n = self.tensors[0].shape[0]
        return (n + self.batch_size - 1) // self.batch_size

    def permutation(self):
        """One epoch's visiting order (int64, on the tensors' device); None = sequential."""
        n = self.tensors[0].shape[0]
        if not (self.shuffle and n):
            return None
        dev = self.tensors[0].device
        perm = torch.empty(n, dtype=torch.int64, device=dev)
        _lib.check(_lib.load().hiprec_random_permutation(_lib.ptr(perm), n, _draw_seed(), _lib.stream_ptr(dev)))
        return perm

    def __iter__(self):
        n = self.tensors[0].shape[0]
        perm = self.permutation()
        for off in range(0, n, self.batch_size):
            if perm is None:
                yield tuple(t[off:off + self.batch_size] for t in self.tensors)
            else:
                idx = perm[off:off + self.batch_size]
                yield tuple(t[idx] for t in self.tensors)


def _train_columns(data, device):
    dev = _resolve_device(device)
    train = data.train
    users = torch.as_tensor(np.asarray(_column(train, DEFAULT_USER_COL))).to(dev, torch.int64)
    items = torch.as_tensor(np.asarray(_column(train, DEFAULT_ITEM_COL))).to(dev, torch.int64)
    return dev, users, items


def instance_bpr_loader(data, batch_size, device, seed=None):
    """base_data.py:218-253: one negative per training row; batches of (user, pos_item, neg_item).
    Returns a :class:`DeviceTripleBatcher` (``user_tensor / pos_item_tensor / neg_item_tensor`` like
    the reference's ``PairwiseNegativeDataset``), which ``MFEngine.train_an_epoch`` runs resident."""
    dev, users, items = _train_columns(data, device)
    neg = sample_negatives(users, items, data.n_users, data.n_items, 1, seed, dev)[:, 0].contiguous()
    print(f"Making PairwiseNegativeDataset of length {users.numel()}")
    return DeviceTripleBatcher(users, items, neg, batch_size, shuffle=True)


def instance_bce_loader(data, batch_size, device, num_negative, seed=None):
    """base_data.py:182-216: every training row (rating kept) followed by ``num_negative`` sampled
    negatives of its user with rating 0; batches of (user, item, rating)."""
    dev, users, items = _train_columns(data, device)
    ratings = torch.as_tensor(np.asarray(_column(data.train, DEFAULT_RATING_COL))).to(dev, torch.float32)
    k = int(num_negative)
    neg = sample_negatives(users, items, data.n_users, data.n_items, k, seed, dev)
    all_users = users.repeat_interleave(k + 1)
    all_items = torch.cat([items[:, None], neg], dim=1).reshape(-1)
    all_ratings = torch.cat([ratings[:, None], torch.zeros_like(neg, dtype=torch.float32)], dim=1).reshape(-1)
    print(f"Making RatingDataset of length {all_users.numel()}")
    return DeviceTensorBatcher((all_users, all_items, all_ratings), batch_size, shuffle=True)


def instance_mul_neg_loader(data, batch_size, device, num_negative, seed=None):
    """base_data.py:254-288: batches of (user, pos_item, neg_items[num_negative])."""
    dev, users, items = _train_columns(data, device)
    neg = sample_negatives(users, items, data.n_users, data.n_items, int(num_negative), seed, dev)
    print(f"Making PairwiseNegativeDataset of length {users.numel()}")
    return DeviceTensorBatcher((users, items, neg), batch_size, shuffle=True)
