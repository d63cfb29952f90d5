"""Drop-in ``MF`` / ``MFEngine`` for beta_rec/models/mf.py, backed by libhiprec.so (HIP, gfx950).

Interface parity with the reference (file:line = /root/reference/beta_rec/...):

* ``MF(config)``            models/mf.py:12-30  — same config keys, same state_dict keys / shapes,
  same initial weights for the same torch seed (the RNG is consumed in the same order).
* ``MF.forward / predict``  models/mf.py:32-70
* ``MFEngine(config)``      models/mf.py:76-90  — incl. quirk Q1 (``reg`` read from the TOP-level
  config, hence 0.0 for every shipped config).
* ``train_single_batch``    models/mf.py:92-119 — returns ``(loss, regularizer)`` python floats.
* ``train_an_epoch``        models/mf.py:121-139 — same print / ``writer.add_scalar`` tags.

What differs is where the arithmetic runs: the gather -> score -> BPR/BCE gradient -> scatter ->
optimizer pipeline is ``csrc/mf.hip`` + ``csrc/optim.hip``; PyTorch only owns the memory.  All five
parameter tensors are views into ONE flat fp32 buffer so that the dense optimizers sweep a single
contiguous range, and so are the gradient accumulator and the optimizer moments.
"""
import ctypes
import time
from functools import wraps

import numpy as np
import torch
import torch.nn as nn
from torch.nn import Parameter

from . import _lib
from .torch_engine import ModelEngine


def timeit(method):
    """Same console contract as beta_rec/utils/common_util.py:215-245."""

    @wraps(method)
    def wrapper(*args, **kw):
        ts = time.time()
        result = method(*args, **kw)
        te = time.time()
        if "log_time" in kw:
            name = kw.get("log_name", method.__name__.upper())
            kw["log_time"][name] = int((te - ts) * 1000)
        else:
            print("Execute [{}] method costing {:2.2f} ms".format(method.__name__, (te - ts) * 1000))
        return result

    return wrapper


class _Table(nn.Module):
    """nn.Embedding look-alike whose ``weight`` is a view into the model's flat buffer."""

    def __init__(self, weight_view):
        super().__init__()
        self.weight = Parameter(weight_view, requires_grad=False)

    @property
    def num_embeddings(self):
        return self.weight.shape[0]

    @property
    def embedding_dim(self):
        return self.weight.shape[1]

    def forward(self, idx):
        """Row gather (bit-exact copy) through hiprec_gather_rows; IndexError like nn.Embedding."""
        return gather_rows(self.weight, idx, check=True)

    def extra_repr(self):
        return f"{self.num_embeddings}, {self.embedding_dim}"


def _new_stats(device, beta1=0.9, beta2=0.999):
    lib = _lib.load()
    stats = torch.zeros(ctypes.sizeof(_lib.Stats), dtype=torch.uint8, device=device)
    _lib.check(lib.hiprec_stats_reset(_lib.ptr(stats), beta1, beta2, _lib.stream_ptr(device)))
    return stats


def read_stats(stats_tensor):
    """Copy the device hiprec_stats block to the host (synchronises) and decode it."""
    raw = stats_tensor.cpu().numpy().tobytes()
    return _lib.Stats.from_buffer_copy(raw)


def clear_status(stats):
    """Zero the 4-byte status word of a device hiprec_stats -- and only that: the struct also holds the optimizer
    clock and the epoch sums, which kernels of other streams may be advancing (writing back a host copy of
    the whole struct raced them)."""
    off = _lib.Stats.status.offset
    stats[off:off + 4].zero_()


def raise_on_status(status):
    """Turn sticky device status bits into the IndexError PyTorch would have raised."""
    if status:
        which = []
        if status & _lib.STATUS_USER_OOB:
            which.append("user")
        if status & _lib.STATUS_ITEM_OOB:
            which.append("item")
        if status & _lib.STATUS_ROW_OOB:
            which.append("row")
        if status & _lib.STATUS_NEG_EXHAUSTED:
            raise ValueError("Sample larger than population or is negative")  # random.sample's message
        if status & _lib.STATUS_LAZY_TABLE:
            raise RuntimeError(
                "lazy Adam: the optimizer ran past the bias-correction table while the corrections still moved "
                "(non-default betas?): use config['model']['dense_opt'] = 'sweep'")
        if status & _lib.STATUS_TABLE_FULL:
            raise RuntimeError(
                "a batch's row-contribution hash partition overflowed while the epoch was staged: the step's lists are "
                "incomplete (should be impossible at >= 4 table entries per triple; please report the batch)")
        if status & _lib.STATUS_ROUTE_OVERFLOW:
            raise RuntimeError(
                "a fixed-capacity all-to-all bucket overflowed: raise the sharded engine's "
                "`route_slack` (config['model']['route_slack'])")
        raise IndexError("index out of range in self (" + "/".join(which) + " index)")


_gather_stats = {}


def gather_rows(table, idx, check=False):
    """out[k] = table[idx[k]] on the GPU, bit-exact (nn.Embedding.forward).  Rows of out-of-range
    indices come back as zeros; with ``check=True`` the call synchronises and raises IndexError for
    them, as nn.Embedding would."""
    if table.device.type != "cuda":
        raise RuntimeError("gather_rows: HIP path only (no CPU fallback)")
    lib = _lib.load()
    dev = table.device
    idx = torch.as_tensor(idx, dtype=torch.int64, device=dev).contiguous()
    flat_idx = idx.reshape(-1)
    out = torch.zeros((flat_idx.numel(), table.shape[1]), dtype=torch.float32, device=dev)
    key = (dev.type, dev.index)
    if key not in _gather_stats:
        _gather_stats[key] = _new_stats(dev)
    stats = _gather_stats[key]
    _lib.check(
        lib.hiprec_gather_rows(
            _lib.ptr(table), table.shape[0], table.shape[1], _lib.ptr(flat_idx),
            flat_idx.numel(), _lib.ptr(out), _lib.ptr(stats), _lib.stream_ptr(dev),
        )
    )
    if check:
        st = read_stats(stats)
        if st.status:
            _gather_stats.pop(key)
            raise_on_status(st.status)
    return out.reshape(*idx.shape, table.shape[1])


class MF(nn.Module):
    """Matrix factorisation model, parameter-compatible with beta_rec/models/mf.py:9-70."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.device = self.config["device_str"]
        self.stddev = self.config["stddev"] if "stddev" in self.config else 0.1
        self.n_users = int(self.config["n_users"])
        self.n_items = int(self.config["n_items"])
        self.emb_dim = int(self.config["emb_dim"])
        U, I, D = self.n_users, self.n_items, self.emb_dim
        self._sizes = (U * D, I * D, U, I, 1)
        flat = torch.empty(sum(self._sizes), dtype=torch.float32)
        ue, ie, ub, ib, gb = self._views(flat)
        # Consume the torch RNG exactly like mf.py:21-30: four nn.Embedding constructors draw
        # N(0,1) for their weights, biases are zero-filled, then the two tables are re-drawn.
        ue.normal_(0, 1)
        ie.normal_(0, 1)
        ub.normal_(0, 1)
        ib.normal_(0, 1)
        ub.fill_(0.0)
        ib.fill_(0.0)
        gb.fill_(0.0)
        ue.normal_(0, self.stddev)
        ie.normal_(0, self.stddev)
        self._flat = flat
        self._table_cache = {}
        self.user_emb = _Table(ue)
        self.item_emb = _Table(ie)
        self.user_bias = _Table(ub)
        self.item_bias = _Table(ib)
        self.global_bias = Parameter(gb, requires_grad=False)
        self._stats = None

    # ---- flat-buffer plumbing -------------------------------------------------------------
    def _views(self, flat):
        U, I, D = self.n_users, self.n_items, self.emb_dim
        o = np.cumsum((0,) + self._sizes)
        return (
            flat[o[0]:o[1]].view(U, D),
            flat[o[1]:o[2]].view(I, D),
            flat[o[2]:o[3]].view(U, 1),
            flat[o[3]:o[4]].view(I, 1),
            flat[o[4]:o[5]],
        )

    def _rebind(self, flat):
        ue, ie, ub, ib, gb = self._views(flat)
        self._flat = flat
        self.user_emb.weight.data = ue
        self.item_emb.weight.data = ie
        self.user_bias.weight.data = ub
        self.item_bias.weight.data = ib
        self.global_bias.data = gb

    def _apply(self, fn, recurse=True):
        """``.to() / .cuda() / .float()``: move the flat buffer once and re-point the views."""
        new_flat = fn(self._flat)
        if new_flat.dtype != torch.float32:
            raise TypeError("hiprec MF keeps fp32 parameters (the reference trains in fp32)")
        if new_flat is not self._flat:
            self._rebind(new_flat.contiguous())
            self._stats = None
        return self

    @property
    def flat(self):
        """The flat fp32 buffer [user_emb | item_emb | user_bias | item_bias | global_bias]."""
        return self._flat

    def tables(self, flat=None):
        """hiprec_mf_tables over this model's layout (``flat`` defaults to the parameters)."""
        flat = self._flat if flat is None else flat
        base = flat.data_ptr()
        cached = self._table_cache.get(base)
        if cached is None:
            o = np.cumsum((0,) + self._sizes) * 4
            cached = _lib.MfTables(
                base + int(o[0]), base + int(o[1]), base + int(o[2]), base + int(o[3]),
                base + int(o[4]), self.n_users, self.n_items, self.emb_dim, 0)
            if len(self._table_cache) > 8:
                self._table_cache.clear()
            self._table_cache[base] = cached
        return cached

    def _require_hip(self):
        if self._flat.device.type != "cuda":
            raise RuntimeError(
                "hiprec MF computes on an MI355X through libhiprec.so only; parameters are on "
                f"{self._flat.device} and there is deliberately no CPU fallback"
            )
        if self._stats is None:
            self._stats = _new_stats(self._flat.device)
        return _lib.load()

    # ---- reference API ---------------------------------------------------------------------
    def forward(self, batch_data):
        """mf.py:32-55 without autograd: ``(sigmoid scores, regularizer)``."""
        users, items = batch_data
        dev = self._flat.device
        users = torch.as_tensor(users, dtype=torch.int64, device=dev).contiguous()
        items = torch.as_tensor(items, dtype=torch.int64, device=dev).contiguous()
        # one launch: the rows are in registers for the dot product, their squared norms ride along (round 4 gathered the
        # four tensors again and reduced each with torch)
        lib, n = self._require_hip(), users.numel()
        if items.numel() != n:
            raise ValueError("users and items must have the same length")
        scores, sq = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
        tabs = self.tables()
        _lib.check(lib.hiprec_mf_forward(ctypes.byref(tabs), _lib.ptr(users), _lib.ptr(items), n, _lib.ptr(scores),
                                         _lib.ptr(sq), _lib.ptr(self._stats), _lib.stream_ptr(dev)))
        # nn.Embedding raises IndexError for an id outside its table (mf.py:41-42); the kernel flags it (NaN score)
        st = read_stats(self._stats)
        if st.status:
            clear_status(self._stats)
            raise_on_status(st.status)
        return scores, sq.sum() / max(n, 1)

    def _scores(self, users, items):
        lib = self._require_hip()
        dev = self._flat.device
        n = users.numel()
        if items.numel() != n:
            raise ValueError("users and items must have the same length")
        scores = torch.empty(n, dtype=torch.float32, device=dev)
        tabs = self.tables()
        _lib.check(
            lib.hiprec_mf_predict(
                ctypes.byref(tabs), _lib.ptr(users), _lib.ptr(items), n, _lib.ptr(scores),
                _lib.ptr(self._stats), _lib.stream_ptr(dev),
            )
        )
        return scores

    def predict(self, users, items):
        """mf.py:57-70: numpy / list ids in, score tensor on the device out."""
        dev = self._flat.device
        users_t = torch.as_tensor(np.asarray(users), dtype=torch.int64).to(dev).contiguous()
        items_t = torch.as_tensor(np.asarray(items), dtype=torch.int64).to(dev).contiguous()
        scores = self._scores(users_t.reshape(-1), items_t.reshape(-1))
        st = read_stats(self._stats)
        if st.status:
            _lib.check(_lib.load().hiprec_stats_reset(_lib.ptr(self._stats), 0.9, 0.999,
                                                      _lib.stream_ptr(dev)))
            raise_on_status(st.status)
        return scores


def sort_within_batches(perm, items, batch_size, n_items):
    """Reorder each batch's slice of the visiting order so that equal items are adjacent.

    Batch composition is unchanged (batch k still holds perm[k*bs:(k+1)*bs] as a set), and the
    loss / gradient of a batch are sums over its triples, so only fp summation order moves.  The
    gradient kernels merge adjacent equal items in LDS (csrc/mf.hip), which removes the same-row
    atomic serialisation that popular (Zipf) items otherwise cause."""
    n = items.numel()
    idx = torch.arange(n, device=items.device) if perm is None else perm
    key = torch.div(torch.arange(n, device=items.device), batch_size, rounding_mode="floor")
    key = key * int(n_items) + items[idx]
    if (n + batch_size - 1) // batch_size * int(n_items) < 2**31:
        key = key.to(torch.int32)  # half the radix passes of the device sort
    return idx[torch.argsort(key)].contiguous()


class PreparedEpoch(tuple):
    """(users, pos, neg, perm, batch_size) of one staged epoch; ``own`` carries the row-ownership arrays of the
    owned-rows SGD step (csrc/mf_owned.hip) when the engine runs that path."""

    own = None


def batch_row_ownership(users, pos, neg, batch_size, n_users, n_items):
    """For an epoch laid out in visiting order: which rows occur ONCE in their batch and which several times
    (``own`` int32 [3, n]: the row's slot inside its batch, or -1; ``total`` int32 [n_batches, stride]:
    occurrences per slot; a row that occurs once has own = -1 or total = 1).  On the device this is ``hiprec_batch_row_ownership`` (one hash table per batch, the
    table position is the slot); :func:`batch_row_ownership_torch` states the same contract with sorts."""
    if users.device.type != "cuda" or n_users + n_items >= 2**31:
        return batch_row_ownership_torch(users, pos, neg, batch_size, n_users, n_items)
    lib = _lib.load()
    n, dev = users.numel(), users.device
    n_batches = max((n + batch_size - 1) // batch_size, 1)
    bits = lib.hiprec_ownership_table_bits(batch_size)
    stride = 1 << bits
    keys = torch.empty(max(lib.hiprec_ownership_ws_ints(n, batch_size, bits), 1), dtype=torch.int32, device=dev)
    total = torch.empty((n_batches, stride), dtype=torch.int32, device=dev)
    own = torch.empty((3, n), dtype=torch.int32, device=dev)
    _lib.check(lib.hiprec_batch_row_ownership(
        _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), n, batch_size, n_users, n_items, bits, _lib.ptr(keys),
        _lib.ptr(total), _lib.ptr(own), _lib.stream_ptr(dev)))
    return own, total, stride


class RowContributions(tuple):
    """(cidx int32 [3, n], rows int32 [n_batches, row_cap, 4], counts int32 [n_batches, 4], row_cap) of a staged
    epoch: hiprec_batch_row_contrib's arrays for the owner-pulls step (hiprec_mf_bpr_epoch_pull)."""


def batch_row_contributions(users, pos, neg, batch_size, n_users, n_items, dim, every_row=False):
    """For an epoch laid out in visiting order: who contributes to which row of a batch (csrc/ownership.hip,
    ``hiprec_batch_row_contrib``).  ``cidx`` -1: the row's only contribution (its contributor updates it in place);
    >= 0: the contribution's place in the step's contribution buffer; -2: a positive occurrence inside its chunk
    neighbour's run.  ``rows`` / ``counts``: the records of the rows with several contributions -- or, ``every_row``
    (the lazy Adam / RMSprop form), of every row of the batch: then no cidx is -1."""
    lib = _lib.load()
    n, dev = users.numel(), users.device
    n_batches = max((n + batch_size - 1) // batch_size, 1)
    bits = lib.hiprec_ownership_table_bits(batch_size)
    min_contrib = 1 if every_row else 2
    row_cap = lib.hiprec_contrib_row_cap(batch_size, min_contrib)
    i32 = dict(dtype=torch.int32, device=dev)
    ws = torch.empty(max(lib.hiprec_ownership_ws_ints(n, batch_size, bits), 1), **i32)
    cidx = torch.empty((3, n), **i32)
    rows = torch.empty((n_batches, row_cap, 4), **i32)
    counts = torch.empty((n_batches, 4), **i32)      # (cleared by the call itself)
    _lib.check(lib.hiprec_batch_row_contrib(
        _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), n, batch_size, n_users, n_items, bits,
        lib.hiprec_mf_pull_chunk(dim), min_contrib, _lib.ptr(ws), _lib.ptr(cidx), _lib.ptr(rows), row_cap, _lib.ptr(counts),
        _lib.stream_ptr(dev)))
    return RowContributions((cidx, rows, counts, row_cap))


def group_epoch_by_item(users, pos, neg, batch_size, n_users, n_items):
    """An epoch in visiting order -> the same batches, each GROUPED BY POSITIVE ITEM (any order of the groups), plus the
    row-ownership arrays of the owned-rows step for the new layout: ``(users, pos, neg, (own, total, stride))``.  No
    sort: ``hiprec_batch_row_ownership_tables`` + ``hiprec_group_epoch_by_item`` (csrc/ownership.hip) -- the hash
    tables the ownership needs anyway are a counting sort by item.  Batch composition is unchanged, so only the fp
    summation order of a step moves (as with :func:`sort_within_batches`, which this replaces on the device)."""
    lib = _lib.load()
    n, dev = users.numel(), users.device
    n_batches = max((n + batch_size - 1) // batch_size, 1)
    bits = lib.hiprec_ownership_table_bits(batch_size)
    stride = 1 << bits
    i32 = dict(dtype=torch.int32, device=dev)
    keys, occ = torch.empty(max(lib.hiprec_ownership_ws_ints(n, batch_size, bits), 1), **i32), torch.empty(3 * n, **i32)
    own, own2 = torch.empty((3, n), **i32), torch.empty((3, n), **i32)
    total, tab_keys, pos_cnt = (torch.empty((n_batches, stride), **i32) for _ in range(3))
    invalid = torch.empty(n_batches, **i32)
    st = _lib.stream_ptr(dev)
    _lib.check(lib.hiprec_batch_row_ownership_tables(
        _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), n, batch_size, n_users, n_items, bits, _lib.ptr(keys),
        _lib.ptr(total), _lib.ptr(own), _lib.ptr(tab_keys), _lib.ptr(pos_cnt), _lib.ptr(occ), st))
    ou, op, on = torch.empty_like(users), torch.empty_like(pos), torch.empty_like(neg)
    _lib.check(lib.hiprec_group_epoch_by_item(
        _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), n, batch_size, n_users, bits, _lib.ptr(own), _lib.ptr(occ),
        _lib.ptr(tab_keys), _lib.ptr(pos_cnt), _lib.ptr(invalid), _lib.ptr(ou), _lib.ptr(op), _lib.ptr(on),
        _lib.ptr(own2), st))
    return ou, op, on, (own2, total, stride)


def batch_row_ownership_torch(users, pos, neg, batch_size, n_users, n_items):
    """For an epoch laid out in visiting order: which rows occur ONCE in their batch and which several times.

    Returns ``(own, total, stride)`` for ``hiprec_mf_bpr_epoch_owned``: ``own`` int32 [3, n] (user / pos / neg
    row of every triple: -1 = the row occurs once in its batch, else its slot id inside the batch), ``total``
    int32 [n_batches, stride] (occurrences per slot; a positive and a negative occurrence of one item both count;
    user rows take the first slots of a batch, item rows follow), ``stride`` = batch_size + batch_size // 2, the
    most slots a batch can need (every slot stands for >= 2 of its 3 * batch_size row occurrences).  Triples with
    an out-of-range id get keys of their own (the kernel skips and flags them).  Integer work on the device (two
    sorts, fixed-size outputs, nothing read back by the host), independent of the weights: it belongs to the
    epoch's staging."""
    n, dev = users.numel(), users.device
    n_batches = (n + batch_size - 1) // batch_size
    stride = batch_size + batch_size // 2
    total = torch.zeros(n_batches * stride + 1, dtype=torch.int32, device=dev)   # last element: dump for non-slots
    if n == 0:
        return torch.zeros((3, 0), dtype=torch.int32, device=dev), total[:-1].view(n_batches, stride), stride
    idx = torch.arange(n, device=dev)
    bid = torch.div(idx, batch_size, rounding_mode="floor")
    ok = ((users >= 0) & (users < n_users) & (pos >= 0) & (pos < n_items) & (neg >= 0) & (neg < n_items))

    def slots(keys, n_rows, first_slot):
        """keys: (batch * n_rows + row) per occurrence, negative = not a real row.  -> slot per occurrence
        (-1 = single) and the number of shared rows per batch; fills `total`."""
        skeys, order = torch.sort(keys)
        new_run = torch.ones_like(skeys, dtype=torch.bool)
        new_run[1:] = skeys[1:] != skeys[:-1]
        run = torch.cumsum(new_run, 0) - 1                                         # run id per sorted position
        cnt = torch.zeros_like(skeys).index_add_(0, run, torch.ones_like(skeys))[run]   # occurrences of its row
        shared = (cnt > 1) & (skeys >= 0)
        head = (shared & new_run).to(torch.int64)                                  # first occurrence of a shared row
        sb = torch.div(skeys, n_rows, rounding_mode="floor").clamp_(min=0)         # batch per sorted position
        n_shared = torch.zeros(n_batches, dtype=torch.int64, device=dev).index_add_(0, sb, head)
        batch_start = torch.cumsum(n_shared, 0) - n_shared                         # shared rows of earlier batches
        slot = torch.cumsum(head, 0) - 1 - batch_start[sb] + first_slot[sb]
        dump = torch.full_like(slot, n_batches * stride)
        total.index_put_((torch.where(head.bool(), sb * stride + slot, dump),), cnt.to(torch.int32))
        out = torch.empty_like(slot)
        out[order] = torch.where(shared, slot, torch.full_like(slot, -1))
        return out, n_shared

    ku = torch.where(ok, bid * n_users + users, -1 - idx)
    su, n_su = slots(ku, n_users, torch.zeros(n_batches, dtype=torch.int64, device=dev))
    ki = torch.cat([torch.where(ok, bid * n_items + pos, -1 - idx), torch.where(ok, bid * n_items + neg, -1 - n - idx)])
    si, _ = slots(ki, n_items, n_su)
    own = torch.stack([su, si[:n], si[n:]]).to(torch.int32).contiguous()
    return own, total[:-1].view(n_batches, stride), stride


class DeviceTripleBatcher:
    """Device-resident replacement for ``DataLoader(PairwiseNegativeDataset, shuffle=True)``.

    beta_rec/data/base_data.py:247-253 builds three int64 tensors and lets the DataLoader index them
    element by element (the reference's end-to-end bottleneck).  Here the arrays
    stay in HBM, a fresh permutation is drawn per epoch and the grad kernel reads its batch through
    ``perm[]`` — no collate.  Iterating it yields ``(users, pos, neg)`` batches like the DataLoader.
    """

    def __init__(self, users, pos_items, neg_items, batch_size, shuffle=True, generator=None):
        self.user_tensor = torch.as_tensor(users, dtype=torch.int64).contiguous()
        dev = self.user_tensor.device
        self.pos_item_tensor = torch.as_tensor(pos_items, dtype=torch.int64, device=dev).contiguous()
        self.neg_item_tensor = torch.as_tensor(neg_items, dtype=torch.int64, device=dev).contiguous()
        if not (len(self.user_tensor) == len(self.pos_item_tensor) == len(self.neg_item_tensor)):
            raise ValueError("users / pos_items / neg_items differ in length")
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.generator = generator

    def __len__(self):
        n = len(self.user_tensor)
        return (n + self.batch_size - 1) // self.batch_size

    def draw_seed(self):
        """Key of one epoch's Feistel shuffle, drawn from torch's global CPU generator."""
        return int(torch.randint(0, 2**62, (1,)).item())

    def native_shuffle(self):
        """True when an epoch's order is P_seed of the device shuffle (no materialised permutation needed)."""
        return self.shuffle and self.generator is None and self.user_tensor.device.type == "cuda"

    def permutation(self):
        """One epoch's visiting order (int64, on the triples' device); None = sequential."""
        if not self.shuffle:
            return None
        n = len(self.user_tensor)
        dev = self.user_tensor.device
        if self.generator is not None:
            return torch.randperm(n, generator=self.generator).to(dev)
        if dev.type != "cuda":
            return torch.randperm(n)
        # native shuffle: a Feistel bijection keyed by a seed drawn from torch's global CPU generator
        # (so torch.manual_seed controls it), no sort
        seed = self.draw_seed()
        perm = torch.empty(n, dtype=torch.int64, device=dev)
        _lib.check(_lib.load().hiprec_random_permutation(_lib.ptr(perm), n, seed, _lib.stream_ptr(dev)))
        return perm

    def __iter__(self):
        perm = self.permutation()
        n = len(self.user_tensor)
        for off in range(0, n, self.batch_size):
            if perm is None:
                sl = slice(off, min(off + self.batch_size, n))
                yield self.user_tensor[sl], self.pos_item_tensor[sl], self.neg_item_tensor[sl]
            else:
                idx = perm[off:off + self.batch_size]
                yield self.user_tensor[idx], self.pos_item_tensor[idx], self.neg_item_tensor[idx]


def _print_config_table(config, tag):
    print("-" * 80)
    print(tag)
    for k, v in config.items():
        print(f"  {k:<16} {v}")
    print("-" * 80)


class MFEngine(ModelEngine):
    """Engine with the surface of beta_rec/models/mf.py:73-139, computing through libhiprec."""

    # SGD keeps every untouched row bit-identical, so it may either sweep the whole flat buffer
    # (cheap while it is cache-resident) or visit only the rows the batch touched.
    ROWS_SGD_MIN_BYTES = 64 << 20
    SORT_MIN_BATCH = 256
    presorted = False  # set when the caller already grouped equal items (sort_within_batches)
    fused_step = True  # resident BPR epochs: one kernel per step (hiprec_mf_bpr_epoch_fused)

    def __init__(self, config):
        self.config = config
        _print_config_table(config["model"], tag="MF model config")
        self.model = MF(config["model"])
        # Quirk Q1 (mf.py:81-83): the key is looked up in the TOP-level config, so this is 0.0 for
        # every shipped config even when config["model"]["reg"] is set.  Reproduced on purpose.
        self.reg = config["model"]["reg"] if "reg" in config else 0.0
        self.batch_size = config["model"]["batch_size"]
        super(MFEngine, self).__init__(config)
        self.model.to(self.device)
        self.loss = self.config["model"]["loss"] if "loss" in self.config["model"] else "bpr"
        print(f"using {self.loss} loss...")
        self._buffers_ready = False
        self._stamp = 1

    # ---- buffers ---------------------------------------------------------------------------
    def _setup(self):
        lib = self.require_hip()
        if self._buffers_ready and self._g_flat.device == self.model.flat.device:
            return lib
        flat = self.model.flat
        dev = flat.device
        self._g_flat = torch.zeros_like(flat)
        self.optimizer.allocate_state(flat)
        self._scratch = torch.zeros(lib.hiprec_scratch_bytes(0), dtype=torch.uint8, device=dev)
        self._stats = _new_stats(dev, self.optimizer.beta1 or 0.9, self.optimizer.beta2 or 0.999)
        # plain SGD on tables beyond the caches visits only the rows of the step.  sgd_mode: "dense" (sweep),
        # "rows" (dense gradient buffer + touched-rows pass, two kernels), "owned" (resident BPR epochs run the
        # one-launch in-place step of csrc/mf_owned.hip, everything else as "rows"), "auto" = dense below 64 MB
        # of parameters, owned above
        rows, self._owned_sgd = self._sgd_modes()
        self._rows_sgd = rows
        if rows:
            self._user_stamp = torch.zeros(self.model.n_users, dtype=torch.int32, device=dev)
            self._item_stamp = torch.zeros(self.model.n_items, dtype=torch.int32, device=dev)
        else:
            self._user_stamp = self._item_stamp = None
        # Adam / RMSprop on tables beyond the caches: `dense_opt` = "sweep" (every element every step, as torch.optim
        # does: 28 bytes per parameter per step), "lazy" (csrc/lazy_opt.hip: resident epochs step only the rows of the
        # batch and replay a lagging row's zero-gradient steps when it is next needed -- bit-identical to the sweep
        # after the flush that ends every epoch) or "auto" (lazy from 64 MB of parameters on)
        mode = self.config["model"].get("dense_opt", "auto")
        if mode not in ("sweep", "lazy", "auto"):
            raise ValueError(f"dense_opt must be 'sweep', 'lazy' or 'auto', not {mode!r}")
        self._lazy = None
        if (self.optimizer.name != "sgd" and self.model.emb_dim <= 256 and self._lazy_capable
                and (mode == "lazy" or (mode == "auto" and flat.numel() * 4 >= self.ROWS_SGD_MIN_BYTES
                                        and _lib.lazy_betas_converge(self.optimizer)))):   # auto: else the sweep
            opt, m = self.optimizer, self.model
            lz = {"stamp_u": torch.full((m.n_users,), -1, dtype=torch.int32, device=dev),
                  "stamp_i": torch.full((m.n_items,), -1, dtype=torch.int32, device=dev),
                  "scalars": _lib.lazy_scalars_table(opt, dev), "dirty": False}   # raises for betas it cannot tabulate
            lz["c"] = _lib.LazyState(
                flat.data_ptr(), self._g_flat.data_ptr(), opt.exp_avg.data_ptr() if opt.exp_avg is not None else None,
                opt.exp_avg_sq.data_ptr(), m.n_users, m.n_items, m.emb_dim, opt.kind, lz["stamp_u"].data_ptr(),
                lz["stamp_i"].data_ptr(), lz["scalars"].data_ptr(), _lib.LAZY_SCALARS_CAP, 0, opt.lr, opt.beta1,
                opt.beta2, opt.eps)
            self._lazy = lz
        self._buffers_ready = True
        return lib

    _lazy_capable = True    # (the data-parallel replicas always sweep: their gradient is dense after the all-reduce)

    def flush_lazy(self):
        """Lazy Adam / RMSprop: replay every lagging row up to the optimizer clock (no-op when nothing lags)."""
        lz = getattr(self, "_lazy", None)
        if lz is not None and lz["dirty"]:
            _lib.check(_lib.load().hiprec_lazy_flush(ctypes.byref(lz["c"]), _lib.ptr(self._stats),
                                                     _lib.stream_ptr(self.model.flat.device)))
            lz["dirty"] = False

    def _lazy_mark_current(self):
        """After anything that moved the clock with a dense sweep (or reset it): every row is current as of it."""
        lz = getattr(self, "_lazy", None)
        if lz is not None:
            _lib.check(_lib.load().hiprec_lazy_mark_current(ctypes.byref(lz["c"]), _lib.ptr(self._stats),
                                                            _lib.stream_ptr(self.model.flat.device)))

    def _take_stamps(self, n):
        if self._stamp + n >= 2**31 - 1:
            self._user_stamp.zero_()
            self._item_stamp.zero_()
            self._stamp = 1
        first = self._stamp
        self._stamp += n
        return first

    def _as_index(self, t):
        return torch.as_tensor(t, device=self.device).to(torch.int64).contiguous()

    # ---- one step ---------------------------------------------------------------------------
    def _prepare_batch(self, batch_data):
        """Validate the loss kind and the batch, move indices to the device (int64, contiguous)."""
        if self.loss == "bpr":
            users, a_items, third = (self._as_index(x) for x in batch_data)
        elif self.loss == "bce":
            users, a_items = (self._as_index(x) for x in batch_data[:2])
            third = torch.as_tensor(batch_data[2], device=self.device).to(torch.float32).contiguous()
        else:
            raise RuntimeError(
                f"Unsupported loss type {self.loss}, try other options: 'bpr' or 'bce'"
            )
        B = users.numel()
        if (B >= self.SORT_MIN_BATCH and not self.presorted and a_items.numel() == B
                and third.numel() == B):
            # group equal (positive) items: the grad kernel merges adjacent duplicates in LDS
            # before touching the dense gradient; a sum over the batch does not depend on order
            order = torch.argsort(a_items)
            users, a_items, third = users[order], a_items[order], third[order]
        if not (a_items.numel() == B and third.numel() == B):
            raise ValueError("batch tensors differ in length")
        if B == 0:
            raise ValueError("empty batch")
        if B == 1:
            # Quirk Q4 (mf.py:44): .squeeze() drops the batch dim of a single-sample batch and
            # torch.sum(dim=1) raises; the reference cannot train on a batch of one.
            raise IndexError("Dimension out of range (expected to be in range of [-1, 0], but got 1)")
        return users, a_items, third

    def _enqueue_grad(self, lib, users, a_items, third):
        """zero_grad + forward + backward: dense gradient of the batch into self._g_flat."""
        m = self.model
        st = _lib.stream_ptr(m.flat.device)
        w, g = m.tables(), m.tables(self._g_flat)
        B = users.numel()
        fn = lib.hiprec_mf_bpr_grad if self.loss == "bpr" else lib.hiprec_mf_bce_grad
        _lib.check(fn(
            ctypes.byref(w), ctypes.byref(g), _lib.ptr(users), _lib.ptr(a_items), _lib.ptr(third),
            None, B, 1.0 / B, float(self.reg), _lib.ptr(self._stats), _lib.ptr(self._scratch),
            self._scratch.numel(), st))

    def _enqueue_step(self, batch_data):
        """Enqueue grad + optimizer kernels for one batch; returns nothing and does not sync."""
        users, a_items, third = self._prepare_batch(batch_data)
        lib = self._setup()
        m = self.model
        st = _lib.stream_ptr(m.flat.device)
        self._enqueue_grad(lib, users, a_items, third)
        opt = self.optimizer
        if self._rows_sgd:
            w, g = m.tables(), m.tables(self._g_flat)
            b_items = third if self.loss == "bpr" else None
            _lib.check(lib.hiprec_mf_sgd_rows(
                ctypes.byref(w), ctypes.byref(g), _lib.ptr(users), _lib.ptr(a_items),
                _lib.ptr(b_items), None, users.numel(), opt.lr, _lib.ptr(self._user_stamp),
                _lib.ptr(self._item_stamp), self._take_stamps(1), _lib.ptr(self._stats),
                _lib.ptr(self._scratch), st))
        else:
            self.flush_lazy()   # (pieces of a lazy epoch may be pending)
            _lib.check(lib.hiprec_opt_dense_step(
                opt.kind, _lib.ptr(m.flat), _lib.ptr(self._g_flat), _lib.ptr(opt.exp_avg),
                _lib.ptr(opt.exp_avg_sq), m.flat.numel(), opt.lr, opt.beta1, opt.beta2, opt.eps,
                _lib.ptr(self._stats), _lib.ptr(self._scratch), m.flat.numel() - 1, st))
            self._lazy_mark_current()

    def backward_only(self, batch_data):
        """zero_grad + forward + backward WITHOUT the optimizer step (what autograd leaves in
        ``p.grad`` in the reference).  Returns ``(loss, regularizer, grads)`` with ``grads`` a dict
        keyed like ``state_dict``; the accumulator is cleared again afterwards.  Note that like
        every grad call it advances the optimizer clock by one."""
        users, a_items, third = self._prepare_batch(batch_data)
        lib = self._setup()
        self._enqueue_grad(lib, users, a_items, third)
        g_global = self.model._views(self._g_flat)[4]
        _lib.check(lib.hiprec_finalize_stats(
            _lib.ptr(self._stats), _lib.ptr(self._scratch), _lib.ptr(g_global), None,
            _lib.stream_ptr(self.model.flat.device)))
        st = self._sync_stats()
        ue, ie, ub, ib, gb = (v.clone() for v in self.model._views(self._g_flat))
        self._g_flat.zero_()
        grads = {"global_bias": gb, "user_emb.weight": ue, "item_emb.weight": ie,
                 "user_bias.weight": ub, "item_bias.weight": ib}
        return st.loss, st.reg, grads

    def load_optimizer_state(self, step, exp_avg=None, exp_avg_sq=None):
        """Restore the optimizer clock and moments (the reference never persists them —
        torch_engine.py:70-73 — so this is an extension used for resume and for tests).
        ``exp_avg`` / ``exp_avg_sq`` are dicts keyed like ``state_dict`` (or None for zeros)."""
        lib = self._setup()
        opt, m = self.optimizer, self.model
        dev = m.flat.device
        _lib.check(lib.hiprec_stats_reset(
            _lib.ptr(self._stats), opt.beta1 or 0.9, opt.beta2 or 0.999, _lib.stream_ptr(dev)))
        _lib.check(lib.hiprec_stats_set_step(_lib.ptr(self._stats), int(step), opt.beta1 or 0.9,
                                             opt.beta2 or 0.999, _lib.stream_ptr(dev)))
        for buf, src in ((opt.exp_avg, exp_avg), (opt.exp_avg_sq, exp_avg_sq)):
            if buf is None:
                continue
            if src is None:
                buf.zero_()
                continue
            ue, ie, ub, ib, gb = m._views(buf)
            for view, key in ((gb, "global_bias"), (ue, "user_emb.weight"), (ie, "item_emb.weight"),
                              (ub, "user_bias.weight"), (ib, "item_bias.weight")):
                view.copy_(torch.as_tensor(src[key], dtype=torch.float32).reshape(view.shape))
        if getattr(self, "_lazy", None) is not None:
            self._lazy["dirty"] = False
            self._lazy_mark_current()   # whatever was loaded is current as of the restored clock

    def optimizer_state(self):
        """(step, exp_avg dict | None, exp_avg_sq dict | None) — counterpart of the loader above."""
        self._setup()
        self.flush_lazy()
        st = read_stats(self._stats)
        out = []
        for buf in (self.optimizer.exp_avg, self.optimizer.exp_avg_sq):
            if buf is None:
                out.append(None)
                continue
            ue, ie, ub, ib, gb = self.model._views(buf)
            out.append({"global_bias": gb.clone(), "user_emb.weight": ue.clone(),
                        "item_emb.weight": ie.clone(), "user_bias.weight": ub.clone(),
                        "item_bias.weight": ib.clone()})
        return st.step, out[0], out[1]

    def _sync_stats(self):
        st = read_stats(self._stats)
        if st.status:
            # clear the sticky bits, keep the optimizer clock
            clear_status(self._stats)
            ob = getattr(self, "_owned_bufs", None)
            if ob is not None:   # a skipped (out-of-range) triple leaves its rows' counts incomplete
                ob["arrived"].zero_()
                ob["acc"].zero_()
            raise_on_status(st.status)
        return st

    def train_single_batch(self, batch_data):
        """mf.py:92-119: one optimisation step; returns ``(loss, regularizer)`` floats."""
        assert hasattr(self, "model"), "Please specify the exact model !"
        self._enqueue_step(batch_data)
        st = self._sync_stats()
        return st.loss, st.reg

    # ---- one epoch --------------------------------------------------------------------------
    def _resident_triples(self, train_loader):
        """(users, pos, neg, perm) when the loader's data is resident and can be batched on device."""
        seed = None
        if isinstance(train_loader, DeviceTripleBatcher):
            if (train_loader.native_shuffle() and train_loader.user_tensor.device == self.model.flat.device
                    and (train_loader.batch_size <= 8192 or self.loss == "bpr")):
                ds, perm, seed = train_loader, None, train_loader.draw_seed()  # shuffle folded into the staging kernel
            else:
                ds, perm = train_loader, train_loader.permutation()
        elif (self.loss == "bce" and len(getattr(train_loader, "tensors", ())) == 3
              and train_loader.tensors[2].is_floating_point() and hasattr(train_loader, "permutation")):
            # data.DeviceTensorBatcher of (user, item, rating): the device-side instance_bce_loader
            class _Rating:
                user_tensor, item_tensor, target_tensor = train_loader.tensors

            ds, perm = _Rating, train_loader.permutation()
        else:
            ds = getattr(train_loader, "dataset", None)
            names = (("user_tensor", "pos_item_tensor", "neg_item_tensor") if self.loss == "bpr"
                     else ("user_tensor", "item_tensor", "target_tensor"))
            if ds is None or not all(hasattr(ds, a) for a in names):
                return None
            if getattr(train_loader, "drop_last", False) or train_loader.batch_sampler is None:
                return None
            # draw the visiting order from the loader's own sampler so that the batches are the
            # ones DataLoader(shuffle=True) would have produced (base_data.py:253); the first draw
            # mirrors _BaseDataLoaderIter's base-seed draw to keep the global RNG in step.
            torch.empty((), dtype=torch.int64).random_()
            order = list(iter(train_loader.sampler))
            perm = torch.as_tensor(order, dtype=torch.int64)
        dev = self.device
        users = ds.user_tensor.to(dev, torch.int64).contiguous()
        if self.loss == "bpr":
            pos = ds.pos_item_tensor.to(dev, torch.int64).contiguous()
            neg = ds.neg_item_tensor.to(dev, torch.int64).contiguous()
        else:  # RatingDataset (data/data_loaders.py:4-27): items + fp32 ratings
            pos = ds.item_tensor.to(dev, torch.int64).contiguous()
            neg = ds.target_tensor.to(dev, torch.float32).contiguous()
        perm = None if perm is None else perm.to(dev).contiguous()
        bs = int(train_loader.batch_size)
        if bs <= 8192 and dev.type == "cuda":
            # native batcher: one block per batch sorts it by item in LDS and writes the epoch in
            # visiting order (hiprec_stage_epoch)
            ou, op, on = torch.empty_like(users), torch.empty_like(pos), torch.empty_like(neg)
            self._stage_into(users, pos, neg, perm, seed, bs, ou, op, on)
            return ou, op, on, None, bs
        if dev.type == "cuda" and self.loss == "bpr" and users.numel() > 0 and bs >= self.SORT_MIN_BATCH:
            # batches beyond the LDS sort (configs[3]: 65 536): sort keys (batch, item) in one launch with the shuffle
            # evaluated on the fly, ONE device sort, one gather of the three arrays through its permutation (round 3:
            # permutation, arange / div / add, argsort and five index gathers -- a dozen launches).  The sorted
            # layout visits neighbouring table rows in neighbouring groups; grouping in hash order instead
            # (group_epoch_by_item: no sort at all) measured 3 % slower per step.
            lib, n = _lib.load(), users.numel()
            n_batches = (n + bs - 1) // bs
            shuffle = 1 if seed is not None else 0
            ws_ints = lib.hiprec_stage_grouped_ws_ints(n, bs, self.model.n_items)
            if ws_ints > 0 and self.config["model"].get("stage_big_batches", "grouped") == "grouped":
                # round 5: a two-level counting sort of this library (ranges of 4096 items, then LDS counters), the
                # gather folded in: three launches, no rocprim, no key / permutation arrays
                ws = self._stage_ws = _lib.grow(getattr(self, "_stage_ws", None), ws_ints, torch.int32, dev)
                ou, op, on = torch.empty_like(users), torch.empty_like(pos), torch.empty_like(neg)
                _lib.check(lib.hiprec_stage_epoch_grouped(
                    _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), _lib.ptr(perm), shuffle, seed or 0, n, bs,
                    self.model.n_items, _lib.ptr(ws), _lib.ptr(ou), _lib.ptr(op), _lib.ptr(on), _lib.stream_ptr(dev)))
                return ou, op, on, None, bs
            small = n_batches * self.model.n_items < 2**31
            keys = torch.empty(n, dtype=torch.int32 if small else torch.int64, device=dev)
            _lib.check(lib.hiprec_stage_sort_keys(_lib.ptr(pos), _lib.ptr(perm), shuffle, seed or 0, n, bs,
                                                  self.model.n_items, 4 if small else 8, _lib.ptr(keys),
                                                  _lib.stream_ptr(dev)))
            order = torch.sort(keys, stable=False).indices
            ou, op, on = torch.empty_like(users), torch.empty_like(pos), torch.empty_like(neg)
            _lib.check(lib.hiprec_gather_epoch(
                _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), _lib.ptr(perm), shuffle, seed or 0, _lib.ptr(order), n,
                _lib.ptr(ou), _lib.ptr(op), _lib.ptr(on), _lib.stream_ptr(dev)))
            return ou, op, on, None, bs
        if seed is not None and perm is None:   # (the sort-free path above did not apply: materialise the shuffle)
            perm = torch.empty(users.numel(), dtype=torch.int64, device=dev)
            _lib.check(_lib.load().hiprec_random_permutation(_lib.ptr(perm), perm.numel(), seed, _lib.stream_ptr(dev)))
        if bs >= self.SORT_MIN_BATCH:
            perm = sort_within_batches(perm, pos, bs, self.model.n_items)
        if perm is not None:
            # lay the epoch out in visiting order (what DataLoader's collate does per batch, done
            # once per epoch on the device): the kernels then read their batch contiguously
            users, pos, neg, perm = users[perm], pos[perm], neg[perm], None
        return users, pos, neg, perm, bs

    def _stage_into(self, users, pos, neg, perm, seed, bs, ou, op, on):
        """hiprec_stage_epoch[_shuffled] on the current stream: one block per batch gathers it in visiting
        order (perm[], P_seed on the fly, or sequential), sorts it by item in LDS and writes it out."""
        lib, dev = _lib.load(), users.device
        if seed is not None:
            _lib.check(lib.hiprec_stage_epoch_shuffled(
                _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), neg.element_size(), seed, users.numel(), bs,
                _lib.ptr(ou), _lib.ptr(op), _lib.ptr(on), _lib.stream_ptr(dev)))
        else:
            _lib.check(lib.hiprec_stage_epoch(
                _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), neg.element_size(), _lib.ptr(perm),
                users.numel(), bs, _lib.ptr(ou), _lib.ptr(op), _lib.ptr(on), _lib.stream_ptr(dev)))

    # ---- next-epoch prefetch ---------------------------------------------------------------------
    # TrainEngine._train (core/train_engine.py:225-240) calls train_an_epoch with the SAME loader every
    # epoch and evaluates in between.  The staging of epoch r+1 (shuffle + per-batch sort + layout) does
    # not depend on the weights, so it runs on a side stream while epoch r trains (it occupies one CU per
    # batch for a few tens of microseconds) into one of two persistent buffer sets.
    def _can_prefetch(self, train_loader):
        return (isinstance(train_loader, DeviceTripleBatcher) and self.loss == "bpr"
                and train_loader.native_shuffle()
                and train_loader.user_tensor.device == self.model.flat.device
                and len(train_loader.user_tensor) > 0)

    def prefetch_epoch(self, train_loader):
        """Stage the NEXT epoch of ``train_loader`` on a side stream.  The following
        :meth:`prepare_epoch` with the same loader object -- and the same, unmodified tensors behind it, see
        :meth:`_loader_fingerprint` -- takes it (after making the current stream wait for it).  Note on the CPU
        generator: the epoch's shuffle seed (``DeviceTripleBatcher.draw_seed``) is drawn when the epoch is STAGED,
        i.e. one epoch earlier than without prefetching, and one extra seed is drawn for the epoch that never runs;
        the sequence of seeds -- and with it the batches of every epoch -- is the same either way.  A prepared epoch obtained this way stays valid until the second prefetch after
        it.  Returns False when the loader cannot be staged ahead."""
        if not self._can_prefetch(train_loader):
            return False
        self._drop_prefetch()
        dev = self.model.flat.device
        main = torch.cuda.current_stream(dev)
        pf = getattr(self, "_pf", None)
        if pf is None or pf["dev"] != dev:
            pf = self._pf = {"dev": dev, "side": torch.cuda.Stream(dev), "slots": [None, None], "next": 0}
        begin = getattr(self, "_ev_epoch_begin", None)
        if begin is not None:
            # everything enqueued before the epoch that is running now (in particular the epoch that last
            # read the slot reused below) is complete when the side stream starts; the running epoch is NOT
            # waited for
            pf["side"].wait_event(begin)
        else:
            pf["side"].wait_stream(main)
        bs = train_loader.batch_size
        if bs <= 8192:
            users, pos, neg = train_loader.user_tensor, train_loader.pos_item_tensor, train_loader.neg_item_tensor
            k = pf["next"]
            pf["next"] = 1 - k
            slot = pf["slots"][k]
            if slot is None or slot[0].numel() != users.numel():
                # allocated with the SIDE stream current: the caching allocator keeps one pool per stream, so
                # these can never be blocks that kernels still in flight on the main stream were just reading
                with torch.cuda.stream(pf["side"]):
                    slot = pf["slots"][k] = tuple(torch.empty_like(t) for t in (users, pos, neg))
            seed = train_loader.draw_seed()
            with torch.cuda.stream(pf["side"]):
                self._stage_into(users, pos, neg, None, seed, bs, *slot)
                prepared = self._finish_staging(slot + (None, bs))
        else:
            # batches beyond the LDS sort: shuffle, per-batch sort and layout are torch ops, enqueued on the
            # side stream like everything else of the staging
            with torch.cuda.stream(pf["side"]):
                prepared = self._finish_staging(self._resident_triples(train_loader))
            for t in prepared[:3]:
                t.record_stream(main)
        with torch.cuda.stream(pf["side"]):
            done = torch.cuda.Event()
            done.record(pf["side"])
        if prepared.own is not None:  # allocated on the side stream, read (and eventually freed) under the main one
            for t in prepared.own:
                if torch.is_tensor(t):
                    t.record_stream(main)
        self._prefetched = (train_loader, done, prepared, self._loader_fingerprint(train_loader))
        return True

    @staticmethod
    def _loader_fingerprint(train_loader):
        """What the staged arrays were made from: address, length and in-place version counter of the loader's
        three tensors and its batch size.  A caller that resamples negatives in place or swaps a tensor between two
        epochs changes it, and the prefetched epoch is dropped instead of training on stale triples."""
        ts = (train_loader.user_tensor, train_loader.pos_item_tensor, train_loader.neg_item_tensor)
        return tuple((t.data_ptr(), t.numel(), t._version) for t in ts) + (int(train_loader.batch_size),)

    def _drop_prefetch(self):
        pending = getattr(self, "_prefetched", None)
        if pending is not None:
            torch.cuda.current_stream(self.model.flat.device).wait_event(pending[1])
            self._prefetched = None

    def prepare_epoch(self, train_loader):
        """Stage one epoch's inputs in HBM: resident triple arrays + this epoch's visiting order.
        Returns an opaque tuple for :meth:`run_prepared_epoch`, or None when the loader's data
        cannot be batched on the device (then train_an_epoch falls back to iterating it)."""
        pending = getattr(self, "_prefetched", None)
        if pending is not None:
            self._prefetched = None
            torch.cuda.current_stream(self.model.flat.device).wait_event(pending[1])
            if pending[0] is train_loader and pending[3] == self._loader_fingerprint(train_loader):
                return pending[2]
        if self.loss not in ("bpr", "bce") or isinstance(train_loader, (list, tuple)):
            return None
        if self.loss == "bce" and isinstance(train_loader, DeviceTripleBatcher):
            return None
        return self._finish_staging(self._resident_triples(train_loader))

    def _sgd_modes(self):
        """(touched-rows SGD?, owned-rows resident step?) -- see _setup."""
        mode = self.config["model"].get("sgd_mode", "auto")
        big = self.model.flat.numel() * 4 >= self.ROWS_SGD_MIN_BYTES
        rows = self.optimizer.name == "sgd" and (mode in ("rows", "owned", "owned_atomic") or (mode == "auto" and big))
        return rows, rows and mode in ("owned", "owned_atomic", "auto") and self.model.emb_dim <= 256

    def _finish_staging(self, staged):
        """Wrap a staged epoch; the owned-rows SGD step also needs to know which rows its batches share."""
        if staged is None:
            return None
        owned = (self._sgd_modes()[1] or self._lazy_owned()) and self.loss == "bpr"
        if getattr(staged, "own", None) is not None:     # the sort-free grouping made the ownership arrays on the way
            if not owned:
                staged.own = None
            return staged
        prepared = PreparedEpoch(staged)
        users, pos, neg, perm, bs = prepared
        if owned and perm is None and users.device.type == "cuda":
            small_keys = self.model.n_users + self.model.n_items < 2**31
            if self._sgd_modes()[1] and self._owned_form() == "pull" and small_keys:
                prepared.own = batch_row_contributions(users, pos, neg, bs, self.model.n_users, self.model.n_items,
                                                       self.model.emb_dim)
            elif self._lazy_owned() == "pull" and small_keys:
                prepared.own = batch_row_contributions(users, pos, neg, bs, self.model.n_users, self.model.n_items,
                                                       self.model.emb_dim, every_row=True)
            else:
                prepared.own = batch_row_ownership(users, pos, neg, bs, self.model.n_users, self.model.n_items)
        return prepared

    def _owned_form(self):
        """``sgd_mode``: "owned" / "auto" = owner pulls (two launches per step, no float atomics: rows several waves
        contribute to are summed from a contribution buffer by one wave each); "owned_atomic" = the one-launch form
        whose shared rows collect device-scope atomic adds (rounds 2-4)."""
        return "atomic" if self.config["model"].get("sgd_mode", "auto") == "owned_atomic" else "pull"

    def _lazy_owned(self):
        """How lazy Adam / RMSprop epochs (BPR) take their gradients -- ``lazy_grad``: "pull" (default, round 5: the
        owner-pulls step of csrc/lazy_opt.hip -- every row's gradient parts through the contribution buffer, ONE launch
        sums, replays the moments and steps a row; dim % 4 == 0, else "owned"), "owned" (round 4: the owned-rows
        kernel writes complete row gradients into the dense buffer, atomics where waves share a row, + the update
        launch) or "atomic" (mf_bpr_grad_kernel's atomics + the update launch).  Returns "pull" / "owned" (truthy: the
        staged epoch carries the matching staging arrays) or False."""
        if self.model.flat.device.type != "cuda" or self.optimizer.name == "sgd":
            return False
        self._setup()
        mode = self.config["model"].get("lazy_grad", "pull")
        if mode not in ("pull", "owned", "atomic"):
            raise ValueError(f"lazy_grad must be 'pull', 'owned' or 'atomic', not {mode!r}")
        if self._lazy is None or mode == "atomic" or self.loss != "bpr":
            return False
        if mode == "pull" and self.model.emb_dim % 4 == 0:
            return "pull"
        return "owned"

    def _fused_ok(self, perm):
        """Cache-sized tables take the one-kernel-per-step epoch driver (any of the three optimizers)."""
        return (self.fused_step and not self._rows_sgd and self.loss == "bpr" and perm is None
                and self.model.emb_dim <= 256)

    def _run_fused_epoch(self, lib, users, pos, neg, n_run, bs, steps=None):
        """hiprec_mf_bpr_epoch_fused: W (and Adam/RMSprop moments) ping-pong between the engine's
        buffers and one alternate each; whatever ends up in the alternates is copied back."""
        m, opt = self.model, self.optimizer
        dev = m.flat.device
        if getattr(self, "_fused_bufs", None) is None or self._fused_bufs["dev"] != dev:
            self._fused_bufs = {
                "dev": dev, "w_alt": torch.empty_like(m.flat),
                "m_alt": torch.empty_like(m.flat) if opt.exp_avg is not None else None,
                "v_alt": torch.empty_like(m.flat) if opt.exp_avg_sq is not None else None,
                "g": [self._g_flat, torch.zeros_like(m.flat), torch.zeros_like(m.flat)],
                "scratch": [torch.zeros_like(self._scratch), torch.zeros_like(self._scratch)],
            }
        fb = self._fused_bufs
        fb["g"][0] = self._g_flat

        def pair(primary, alt):
            return None if primary is None else (ctypes.c_void_p * 2)(primary.data_ptr(), alt.data_ptr())

        w_arr = pair(m.flat, fb["w_alt"])
        m_arr = pair(opt.exp_avg, fb["m_alt"])
        v_arr = pair(opt.exp_avg_sq, fb["v_alt"])
        g_arr = (ctypes.c_void_p * 3)(*(t.data_ptr() for t in fb["g"]))
        s_arr = (ctypes.c_void_p * 2)(*(t.data_ptr() for t in fb["scratch"]))
        final = ctypes.c_int32(-1)
        n_steps = (n_run + bs - 1) // bs
        a, b = (0, n_steps) if steps is None else steps
        _lib.check(lib.hiprec_mf_bpr_epoch_fused_range(
            opt.kind, w_arr, g_arr, m_arr, v_arr, s_arr, m.n_users, m.n_items, m.emb_dim,
            _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), n_run, bs, a, b, float(self.reg), opt.lr,
            opt.beta1, opt.beta2, opt.eps, _lib.ptr(self._stats), ctypes.byref(final),
            _lib.stream_ptr(dev)))
        if final.value == 1:  # the state ended up in the alternate buffers
            m.flat.copy_(fb["w_alt"])
            if opt.exp_avg is not None:
                opt.exp_avg.copy_(fb["m_alt"])
            if opt.exp_avg_sq is not None:
                opt.exp_avg_sq.copy_(fb["v_alt"])

    def run_prepared_epoch(self, prepared, sync=True, prefetch=None, steps=None):
        """Enqueue every step of a prepared epoch (hiprec_mf_bpr_epoch, or the fused one-kernel-per-
        step driver).  With ``sync=False`` nothing is read back; call :meth:`epoch_stats` later.
        ``prefetch`` = the loader whose NEXT epoch is to be staged on the side stream meanwhile
        (:meth:`prefetch_epoch`; issued before this epoch's launches so that the host does not hand it
        to the GPU only when the epoch is nearly over).  ``steps=(a, b)`` enqueues only steps [a, b) of
        the epoch (consecutive calls with the same ``prepared`` cover it piece by piece, e.g. to put a
        timestamp in between; the weights are valid again once a piece reaches the last step)."""
        lib = self._setup()
        users, pos, neg, perm, bs = prepared
        n = users.numel()
        n_run = n - 1 if n % bs == 1 else n  # Q4: a trailing batch of one raises (below)
        if steps is not None:
            n_steps = (n_run + bs - 1) // bs
            if not (0 <= steps[0] <= steps[1] <= n_steps):
                raise ValueError(f"steps {steps} outside the epoch's {n_steps} steps")
            if sync:
                raise ValueError("a partial epoch cannot be read back: pass sync=False")
        if steps is None or steps[0] == 0:
            self._begin_epoch_marker(prepared, prefetch)
        if self._owned_sgd and self.loss == "bpr" and getattr(prepared, "own", None) is not None:
            self._run_owned_epoch(lib, prepared, n_run, steps)
            if not sync:
                return None
            st = self._sync_stats()
            if n_run != n:
                raise IndexError(
                    "Dimension out of range (expected to be in range of [-1, 0], but got 1)")
            return st
        if self._lazy is not None and perm is None and self.loss in ("bpr", "bce"):
            self._run_lazy_epoch(lib, users, pos, neg, n_run, bs, steps, getattr(prepared, "own", None))
            if not sync:
                return None
            st = self._sync_stats()
            if n_run != n:
                raise IndexError(
                    "Dimension out of range (expected to be in range of [-1, 0], but got 1)")
            return st
        if self._fused_ok(perm):
            self._run_fused_epoch(lib, users, pos, neg, n_run, bs, steps)
            if not sync:
                return None
            st = self._sync_stats()
            if n_run != n:
                raise IndexError(
                    "Dimension out of range (expected to be in range of [-1, 0], but got 1)")
            return st
        if steps is not None:
            # the launch-per-kernel drivers keep no state between steps: a piece is a slice of the arrays
            a, b = steps[0] * bs, min(steps[1] * bs, n_run)
            users, pos, neg, n_run = users[a:b], pos[a:b], neg[a:b], b - a
            perm = None if perm is None else perm[a:b]
        return self._run_unfused_epoch(lib, users, pos, neg, perm, bs, n, n_run, sync)

    def _run_lazy_epoch(self, lib, users, items_a, third, n_run, bs, steps, own=None):
        """hiprec_mf_epoch_lazy (csrc/lazy_opt.hip): per step catch-up of the batch's rows, the gradient kernel, the
        update of the batch's rows; the piece that reaches the last step flushes, so the tables and the moments hold
        what torch.optim's dense steps would have left whenever an epoch is over."""
        m, lz = self.model, self._lazy
        n_steps = (n_run + bs - 1) // bs
        a, b = (0, n_steps) if steps is None else steps
        lo, hi = a * bs, min(b * bs, n_run)
        w, g = m.tables(), m.tables(self._g_flat)
        el = third.element_size()
        lz["dirty"] = True
        if isinstance(own, RowContributions) and self.loss == "bpr" and self._lazy_owned() == "pull":
            cidx, rows, counts, row_cap = own
            n_all = cidx.shape[1]
            cap = 3 * min(bs, max(n_all, 1))
            pb = getattr(self, "_pull_bufs", None)
            if pb is None or pb["dev"] != m.flat.device or pb["cap"] < cap:
                pb = self._pull_bufs = {
                    "dev": m.flat.device, "cap": cap,
                    "cbuf": torch.empty(cap * m.emb_dim, dtype=torch.float32, device=m.flat.device),
                    "cbias": torch.empty(cap, dtype=torch.float32, device=m.flat.device),
                    "scratch": torch.zeros_like(self._scratch)}
            _lib.check(lib.hiprec_mf_epoch_lazy_pull(
                ctypes.byref(lz["c"]), ctypes.c_void_p(users.data_ptr() + 8 * lo),
                ctypes.c_void_p(items_a.data_ptr() + 8 * lo), ctypes.c_void_p(third.data_ptr() + 8 * lo),
                ctypes.c_void_p(cidx.data_ptr() + 4 * lo), n_all, ctypes.c_void_p(rows.data_ptr() + 16 * a * row_cap),
                row_cap, ctypes.c_void_p(counts.data_ptr() + 16 * a), _lib.ptr(pb["cbuf"]), _lib.ptr(pb["cbias"]),
                hi - lo, bs, 1 if a == 0 else 0, float(self.reg), _lib.ptr(self._stats), _lib.ptr(self._scratch),
                _lib.stream_ptr(m.flat.device)))
            if b == n_steps:
                self.flush_lazy()
            return
        if own is not None and not isinstance(own, RowContributions) and self.loss == "bpr" and self._lazy_owned():
            # the owned-rows gradient kernel: the epoch's ownership arrays, sliced like the triples
            o, total, stride = own
            _lib.check(lib.hiprec_mf_epoch_lazy_owned(
                ctypes.byref(lz["c"]), ctypes.c_void_p(users.data_ptr() + 8 * lo),
                ctypes.c_void_p(items_a.data_ptr() + 8 * lo), ctypes.c_void_p(third.data_ptr() + 8 * lo),
                ctypes.c_void_p(o[0].data_ptr() + 4 * lo), ctypes.c_void_p(o[1].data_ptr() + 4 * lo),
                ctypes.c_void_p(o[2].data_ptr() + 4 * lo), ctypes.c_void_p(total.data_ptr() + 4 * a * stride), stride,
                hi - lo, bs, 1 if a == 0 else 0, float(self.reg), _lib.ptr(self._stats), _lib.ptr(self._scratch),
                _lib.stream_ptr(m.flat.device)))
            if b == n_steps:
                self.flush_lazy()
            return
        _lib.check(lib.hiprec_mf_epoch_lazy(
            ctypes.byref(lz["c"]), ctypes.byref(w), ctypes.byref(g), ctypes.c_void_p(users.data_ptr() + 8 * lo),
            ctypes.c_void_p(items_a.data_ptr() + 8 * lo), ctypes.c_void_p(third.data_ptr() + el * lo),
            0 if self.loss == "bpr" else 1, hi - lo, bs, 1 if a == 0 else 0, float(self.reg), _lib.ptr(self._stats),
            _lib.ptr(self._scratch), self._scratch.numel(), _lib.stream_ptr(m.flat.device)))
        if b == n_steps:
            self.flush_lazy()

    def _run_owned_epoch(self, lib, prepared, n_run, steps):
        """hiprec_mf_bpr_epoch_owned (csrc/mf_owned.hip): plain SGD on tables beyond the caches, one launch per
        step, rows updated in place by the wave that holds their complete gradient.  Work space: one compact
        accumulator row + arrival counter per shared row of a batch (zero between steps), nothing table-sized."""
        users, pos, neg, _, bs = prepared
        if isinstance(prepared.own, RowContributions):
            return self._run_pull_epoch(lib, prepared, n_run, steps)
        own, total, stride = prepared.own
        m = self.model
        dev = m.flat.device
        ob = getattr(self, "_owned_bufs", None)
        if ob is None or ob["dev"] != dev or ob["stride"] < stride:
            ob = self._owned_bufs = {
                "dev": dev, "stride": stride,
                "arrived": torch.zeros(stride, dtype=torch.int32, device=dev),
                "acc": torch.zeros(stride * (m.emb_dim + 1), dtype=torch.float32, device=dev),
                "gb": torch.zeros(2, dtype=torch.float32, device=dev),
                "scratch": [torch.zeros_like(self._scratch), torch.zeros_like(self._scratch)]}
        s_arr = (ctypes.c_void_p * 2)(*(t.data_ptr() for t in ob["scratch"]))
        n_steps = (n_run + bs - 1) // bs
        a, b = (0, n_steps) if steps is None else steps
        _lib.check(lib.hiprec_mf_bpr_epoch_owned(
            _lib.ptr(m.flat), m.n_users, m.n_items, m.emb_dim, _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg),
            _lib.ptr(own[0]), _lib.ptr(own[1]), _lib.ptr(own[2]), _lib.ptr(total), stride,
            _lib.ptr(ob["arrived"]), _lib.ptr(ob["acc"]), ob["stride"], _lib.ptr(ob["gb"]), s_arr, n_run, bs,
            a, b, float(self.reg), self.optimizer.lr, _lib.ptr(self._stats), _lib.stream_ptr(dev)))

    def _run_pull_epoch(self, lib, prepared, n_run, steps):
        """hiprec_mf_bpr_epoch_pull (csrc/mf_owned.hip): the owned-rows step as owner pulls.  Work space: one
        contribution row per row occurrence of a batch (worst case; plain stores, never cleared)."""
        users, pos, neg, _, bs = prepared
        cidx, rows, counts, row_cap = prepared.own
        m = self.model
        dev = m.flat.device
        n = users.numel()
        cap = 3 * min(bs, max(n, 1))
        pb = getattr(self, "_pull_bufs", None)
        if pb is None or pb["dev"] != dev or pb["cap"] < cap:
            pb = self._pull_bufs = {
                "dev": dev, "cap": cap,
                "cbuf": torch.empty(cap * m.emb_dim, dtype=torch.float32, device=dev),
                "cbias": torch.empty(cap, dtype=torch.float32, device=dev),
                "scratch": torch.zeros_like(self._scratch)}
        n_steps = (n_run + bs - 1) // bs
        a, b = (0, n_steps) if steps is None else steps
        _lib.check(lib.hiprec_mf_bpr_epoch_pull(
            _lib.ptr(m.flat), m.n_users, m.n_items, m.emb_dim, _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg),
            _lib.ptr(cidx), n, _lib.ptr(rows), row_cap, _lib.ptr(counts), _lib.ptr(pb["cbuf"]), _lib.ptr(pb["cbias"]),
            _lib.ptr(pb["scratch"]), n_run, bs, a, b, float(self.reg), self.optimizer.lr, _lib.ptr(self._stats),
            _lib.stream_ptr(dev)))

    def _begin_epoch_marker(self, prepared, prefetch):
        self._ev_epoch_begin = torch.cuda.Event()
        self._ev_epoch_begin.record(torch.cuda.current_stream(self.model.flat.device))
        # the staged arrays must outlive the kernels that read them: a caller's temporary would be freed as soon
        # as this call returns, with the epoch still in flight (fine for later work on THIS stream, not for the
        # side-stream staging of the next epoch).  Held until the next epoch replaces them.
        self._epoch_inputs = prepared
        if prefetch is not None:
            self.prefetch_epoch(prefetch)

    def _run_unfused_epoch(self, lib, users, pos, neg, perm, bs, n, n_run, sync):
        m, opt = self.model, self.optimizer
        w, g = m.tables(), m.tables(self._g_flat)
        n_batches = (n_run + bs - 1) // bs
        first = self._take_stamps(n_batches) if self._rows_sgd else 0
        epoch_fn = lib.hiprec_mf_bpr_epoch if self.loss == "bpr" else lib.hiprec_mf_bce_epoch
        _lib.check(epoch_fn(
            ctypes.byref(w), ctypes.byref(g), _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg),
            _lib.ptr(perm), n_run, bs, float(self.reg), opt.kind, opt.lr, opt.beta1, opt.beta2,
            opt.eps, _lib.ptr(m.flat), _lib.ptr(self._g_flat), _lib.ptr(opt.exp_avg),
            _lib.ptr(opt.exp_avg_sq), m.flat.numel(), _lib.ptr(self._user_stamp),
            _lib.ptr(self._item_stamp), first, _lib.ptr(self._stats),
            _lib.ptr(self._scratch), self._scratch.numel(), _lib.stream_ptr(m.flat.device)))
        if not sync:
            return None
        st = self._sync_stats()
        if n_run != n:
            raise IndexError(
                "Dimension out of range (expected to be in range of [-1, 0], but got 1)")
        return st

    def epoch_stats(self):
        """Synchronise and return the device statistics (loss, reg, loss_sum, reg_sum, step)."""
        return self._sync_stats()

    @timeit
    def train_an_epoch(self, train_loader, epoch_id):
        """mf.py:121-139.  One host sync per epoch instead of two per step."""
        assert hasattr(self, "model"), "Please specify the exact model !"
        self.model.train()
        lib = self._setup()
        prepared = self.prepare_epoch(train_loader)
        if prepared is not None:
            # the next epoch's shuffle + staging overlaps this one (side stream)
            ahead = train_loader if self.config["model"].get("prefetch_epoch", True) else None
            self.run_prepared_epoch(prepared, sync=False, prefetch=ahead)
            n, bs = prepared[0].numel(), prepared[4]
            st = self._sync_stats()
            if n % bs == 1:  # Q4: a trailing batch of one raises in the reference
                raise IndexError("Dimension out of range (expected to be in range of [-1, 0], but got 1)")
        else:
            _lib.check(lib.hiprec_stats_begin_epoch(
                _lib.ptr(self._stats), _lib.stream_ptr(self.model.flat.device)))
            for batch_data in train_loader:
                self._enqueue_step(batch_data)
            st = self._sync_stats()
        loss, total_loss, regularizer = st.loss, st.loss_sum, st.reg_sum
        print(f"[Training Epoch {epoch_id}], Loss {loss}, Regularizer {regularizer}")
        self.writer.add_scalar("model/loss", total_loss, epoch_id)
        self.writer.add_scalar("model/regularizer", regularizer, epoch_id)
