"""Drop-in ``Triple2vec`` / ``Triple2vecEngine`` for beta_rec/models/triple2vec.py on libhiprec.so.

SURVEY.md §8f rank 4 ("sibling models on the same kernels").  Interface parity (file:line =
/root/reference/beta_rec/...): ``Triple2vec(config)`` models/triple2vec.py:8-104 (``forward(batch) ->
loss``, ``predict(users, items)``), ``Triple2vecEngine(config)`` :107-169 (``train_single_batch(batch)
-> float``, ``train_an_epoch(loader, epoch_id)`` which draws the negatives from ``engine.data``'s alias
samplers).  Same config keys (``n_users n_items emb_dim n_neg batch_size optimizer lr device_str``),
same ``state_dict`` keys, same initial weights for the same torch seed.

Quirks kept on purpose (the oracle pins them against the real reference):
* ``use_bias = n_neg`` (triple2vec.py:19): for every usable config ``item_emb2`` becomes the SAME
  module as ``item_emb1`` on the first forward (triple2vec.py:38-39); the orphaned table never
  receives a gradient again.  Here the shared table is passed twice to the kernels and the orphan is
  left out of the optimizer sweep.
* both negative item rows are gathered with ``neg_i_2``; ``neg_i_1`` only selects an ``item_bias``
  entry (triple2vec.py:46-47, 69-71).
* the loss is divided by ``3 * config batch_size`` also for a short last batch (triple2vec.py:92).

Loss, backward, predict and (optionally) the alias sampling run in ``csrc/triple2vec.hip``; the
optimizer sweep is the shared ``csrc/optim.hip``.  There is no CPU path.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .mf import _new_stats, raise_on_status, read_stats
from .ncf import _FlatModel, _ParamView
from .flat_engine import FlatModelEngine


class Triple2vec(_FlatModel):
    """models/triple2vec.py:8-104.  Flat buffer: [user_emb | item_emb1 | user_bias | item_bias | item_emb2]
    — the table that gets orphaned sits last so that the optimizer sweep can stop in front of it."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.n_users = int(config["n_users"])
        self.n_items = int(config["n_items"])
        self.emb_dim = int(config["emb_dim"])
        self.n_neg = config["n_neg"]
        self.use_bias = config["n_neg"]          # sic, triple2vec.py:19
        self.batch_size = config["batch_size"]
        U, I, D = self.n_users, self.n_items, self.emb_dim
        v = self._build([("user_emb.weight", (U, D)), ("item_emb1.weight", (I, D)),
                         ("user_bias.weight", (U, 1)), ("item_bias.weight", (I, 1)),
                         ("item_emb2.weight", (I, D))])
        # RNG order of triple2vec.py:21-34: five nn.Embedding (N(0,1) each), then uniform_ x3, fill_ x2
        for name in ("user_emb.weight", "item_emb1.weight", "item_emb2.weight", "user_bias.weight",
                     "item_bias.weight"):
            v[name].normal_(0, 1)
        for name in ("user_emb.weight", "item_emb1.weight", "item_emb2.weight"):
            v[name].uniform_(-0.01, 0.01)
        v["user_bias.weight"].fill_(0.0)
        v["item_bias.weight"].fill_(0.0)
        # registration order = the reference's state_dict order
        self.user_emb = _ParamView(v["user_emb.weight"])
        self.item_emb1 = _ParamView(v["item_emb1.weight"])
        self.item_emb2 = _ParamView(v["item_emb2.weight"])
        self.user_bias = _ParamView(v["user_bias.weight"])
        self.item_bias = _ParamView(v["item_bias.weight"])
        self._stats = None

    # ---- the item_emb2 alias ------------------------------------------------------------------------
    @property
    def shared_items(self):
        return self.item_emb2 is self.item_emb1

    def _alias(self):
        """triple2vec.py:38-39."""
        if self.use_bias and not self.shared_items:
            self.item_emb2 = self.item_emb1

    def _rebind(self, flat):
        self._flat = flat
        for name, view in self.views(flat).items():
            if name == "item_emb2.weight" and self.shared_items:
                continue                         # the name now resolves to item_emb1's parameter
            mod, attr = self._owner(name)
            getattr(mod, attr).data = view

    def n_active(self):
        """Floats the optimizer moves: everything in front of the orphaned item_emb2 once it is aliased."""
        return self.offset_of("item_emb2.weight") if self.shared_items else self._flat.numel()

    def tables(self, flat=None):
        """hiprec_t2v_tables over the weight buffer (or a same-shaped gradient buffer)."""
        flat = self._flat if flat is None else flat
        at = lambda name: flat.data_ptr() + 4 * self.offset_of(name)  # noqa: E731
        e1 = at("item_emb1.weight")
        e2 = e1 if self.shared_items else at("item_emb2.weight")
        return _lib.T2vTables(at("user_emb.weight"), e1, e2, at("user_bias.weight"), at("item_bias.weight"),
                              self.n_users, self.n_items, self.emb_dim, 0)

    def _device_stats(self):
        dev = self._flat.device
        if self._stats is None or self._stats.device != dev:
            self._stats = _new_stats(dev)
        return self._stats

    def _check_status(self):
        s = read_stats(self._stats)
        if s.status:
            self._stats = None
            raise_on_status(s.status)
        return s

    # ---- reference API -------------------------------------------------------------------------------
    def forward(self, batch_data):
        """triple2vec.py:36-92: the batch loss as a 0-dim tensor (no autograd graph: training goes
        through ``Triple2vecEngine.train_single_batch``, which keeps the gradient this call discards)."""
        lib = self._require_hip()
        self._alias()
        dev = self._flat.device
        stats = self._device_stats()
        g = torch.zeros_like(self._flat)
        scratch = torch.zeros(lib.hiprec_scratch_bytes(0), dtype=torch.uint8, device=dev)
        enqueue_grad(lib, self, g, batch_data, stats, scratch)
        _lib.check(lib.hiprec_finalize_stats(_lib.ptr(stats), _lib.ptr(scratch), None, None, _lib.stream_ptr(dev)))
        return torch.tensor(self._check_status().loss, device=dev)

    def predict(self, users, items):
        """triple2vec.py:94-104."""
        lib = self._require_hip()
        dev = self._flat.device
        users_t, items_t = (x.to(dev, torch.int64).reshape(-1).contiguous() if torch.is_tensor(x) else
                            torch.as_tensor(np.asarray(x), dtype=torch.int64).to(dev).reshape(-1).contiguous()
                            for x in (users, items))
        if users_t.numel() != items_t.numel():
            raise ValueError("users and items differ in length")
        stats = self._device_stats()
        scores = torch.empty(users_t.numel(), dtype=torch.float32, device=dev)
        w = self.tables()
        _lib.check(lib.hiprec_t2v_predict(ctypes.byref(w), _lib.ptr(users_t), _lib.ptr(items_t), users_t.numel(),
                                          _lib.ptr(scores), _lib.ptr(stats), _lib.stream_ptr(dev)))
        self._check_status()
        return scores


def _batch_tensors(model, batch_data):
    """(pos_u, pos_i_1, pos_i_2, neg_u, neg_i_1, neg_i_2) -> contiguous int64 device tensors, n_neg."""
    if len(batch_data) != 6:
        raise ValueError("Triple2vec batches are (pos_u, pos_i_1, pos_i_2, neg_u, neg_i_1, neg_i_2)")
    dev = model.flat.device
    t = [torch.as_tensor(x, device=dev).to(torch.int64).contiguous() for x in batch_data]
    B = t[0].numel()
    if B == 0:
        raise ValueError("empty batch")
    if not (t[1].numel() == B and t[2].numel() == B):
        raise ValueError("positive index tensors differ in length")
    n_neg = t[3].numel() // B
    if not all(x.numel() == B * n_neg for x in t[3:]):
        raise ValueError("negative index tensors must be [batch, n_neg]")
    return t, B, n_neg


def enqueue_grad(lib, model, g_flat, batch_data, stats, scratch):
    t, B, n_neg = _batch_tensors(model, batch_data)
    w, g = model.tables(), model.tables(g_flat)
    _lib.check(lib.hiprec_t2v_grad(
        ctypes.byref(w), ctypes.byref(g), *(_lib.ptr(x) for x in t), B, n_neg,
        1.0 / (3 * model.batch_size), _lib.ptr(stats), _lib.ptr(scratch), scratch.numel(),
        _lib.stream_ptr(model.flat.device)))


class DeviceAliasTable:
    """``AliasTable`` (utils/alias_table.py) resident on the device: ``sample(count, obj_num)`` returns an
    ``[obj_num, count]`` int64 device tensor drawn by ``hiprec_alias_sample`` instead of a python loop."""

    def __init__(self, table, device):
        self.vocab = len(table.prob_arr)
        self.prob = torch.as_tensor(np.asarray(table.prob_arr, dtype=np.float64)).to(device)
        self.alias = torch.as_tensor(np.asarray(table.alias_arr, dtype=np.int64)).to(device)
        labels = np.asarray(table.index2Label, dtype=np.int64)
        self.labels = None if np.array_equal(labels, np.arange(self.vocab)) else torch.as_tensor(labels).to(device)
        self.device = device

    def sample(self, count, obj_num, seed):
        lib = _lib.load()
        out = torch.empty(obj_num, count, dtype=torch.int64, device=self.device)
        _lib.check(lib.hiprec_alias_sample(_lib.ptr(self.prob), _lib.ptr(self.alias), _lib.ptr(self.labels),
                                           self.vocab, int(seed) & (2 ** 64 - 1), _lib.ptr(out), out.numel(),
                                           _lib.stream_ptr(self.device)))
        return out


class Triple2vecEngine(FlatModelEngine):
    """models/triple2vec.py:107-169."""

    def __init__(self, config):
        self.config = config
        self.model = Triple2vec(config["model"])
        super(Triple2vecEngine, self).__init__(config)
        m = config["model"]
        # "host": engine.data.{user,item}_sampler.sample(...) exactly as triple2vec.py:145-166 (same
        # numpy random stream as the reference); "device": the same alias tables sampled on the GPU
        self.negative_sampler = m["negative_sampler"] if "negative_sampler" in m else "host"
        self.sampler_seed = int(m["sampler_seed"]) if "sampler_seed" in m else 0
        self._device_samplers = None

    def _sweep_floats(self):
        return self.model.n_active()      # the orphaned item_emb2 sits last and stays out of the sweep

    def _enqueue_grad(self, batch_data):
        lib = self._setup()
        self.model._alias()
        enqueue_grad(lib, self.model, self._g_flat, batch_data, self._stats, self._scratch)

    def train_single_batch(self, batch_data, ratings=None):
        """triple2vec.py:115-124: one step, returns ``loss.item()``."""
        assert hasattr(self, "model"), "Please specify the exact model !"
        self._enqueue_step(batch_data)
        return self._sync_stats().loss

    # ---- epoch ---------------------------------------------------------------------------------------
    def _negatives(self, n, epoch_id, batch_id):
        """Three [n, n_neg] draws: users, items, items (triple2vec.py:145-166)."""
        n_neg = self.config["model"]["n_neg"]
        dev = self.model.flat.device
        if self.negative_sampler == "host":
            return [torch.tensor(s.sample(n_neg, n), dtype=torch.int64, device=dev).reshape(n, -1)
                    for s in (self.data.user_sampler, self.data.item_sampler, self.data.item_sampler)]
        if self.negative_sampler != "device":
            raise ValueError(f"unknown negative_sampler {self.negative_sampler!r}: 'host' or 'device'")
        if self._device_samplers is None or self._device_samplers[0].device != dev:
            self._device_samplers = (DeviceAliasTable(self.data.user_sampler, dev),
                                     DeviceAliasTable(self.data.item_sampler, dev))
        us, its = self._device_samplers
        base = (self.sampler_seed * 1_000_003 + int(epoch_id)) * 1_000_003 + batch_id
        return [us.sample(n_neg, n, 3 * base), its.sample(n_neg, n, 3 * base + 1), its.sample(n_neg, n, 3 * base + 2)]

    def enqueue_epoch(self, pos_u, pos_i1, pos_i2, neg_u, neg_i1, neg_i2):
        """One epoch over resident device arrays in visiting order (``pos_*[N]``, ``neg_*[N, n_neg]``;
        batches of the configured ``batch_size``, the last one short), enqueued by ``hiprec_t2v_epoch``
        with no host work between steps and no sync."""
        lib = self._setup()
        m, opt = self.model, self.optimizer
        m._alias()
        dev = m.flat.device
        t = [x.to(dev).to(torch.int64).contiguous() for x in (pos_u, pos_i1, pos_i2, neg_u, neg_i1, neg_i2)]
        N = t[0].numel()
        if N == 0:
            raise ValueError("empty epoch")
        n_neg = t[3].numel() // N
        if not (t[1].numel() == N and t[2].numel() == N and all(x.numel() == N * n_neg for x in t[3:])):
            raise ValueError("epoch arrays must be pos_*[N] and neg_*[N, n_neg]")
        w, g = m.tables(), m.tables(self._g_flat)
        _lib.check(lib.hiprec_t2v_epoch(
            ctypes.byref(w), ctypes.byref(g), *(_lib.ptr(x) for x in t), N, int(m.batch_size), n_neg,
            1.0 / (3 * m.batch_size), opt.kind, opt.lr, opt.beta1, opt.beta2, opt.eps, _lib.ptr(m.flat),
            _lib.ptr(self._g_flat), _lib.ptr(opt.exp_avg), _lib.ptr(opt.exp_avg_sq), m.n_active(),
            _lib.ptr(self._stats), _lib.ptr(self._scratch), self._scratch.numel(), _lib.stream_ptr(dev)))

    def train_an_epoch(self, train_loader, epoch_id):
        """triple2vec.py:126-169: every ``sample`` of the loader is a ``[B, 3]`` block of (u, i1, i2)
        triples; negatives are drawn per batch (host sampler, in the reference's order: users, items,
        items) or for the whole epoch at once (device sampler); prints the LAST batch's loss and logs the
        epoch sum.  The epoch is enqueued from C with one host sync at the end."""
        assert hasattr(self, "model"), "Please specify the exact model !"
        self.model.train()
        self._setup()
        dev = self.model.flat.device
        blocks, negs = [], []
        for batch_id, sample in enumerate(train_loader):
            sample = torch.as_tensor(sample, device=dev).to(torch.int64).reshape(-1, 3)
            blocks.append(sample)
            if self.negative_sampler == "host":
                negs.append(self._negatives(sample.shape[0], epoch_id, batch_id))
        if not blocks:
            raise ValueError("empty epoch")
        B = self.model.batch_size
        if any(b.shape[0] != B for b in blocks[:-1]) or blocks[-1].shape[0] > B:
            raise ValueError("every batch but the last must hold batch_size triples")
        epoch = torch.cat(blocks)
        if self.negative_sampler == "host":
            neg = [torch.cat([n[k] for n in negs]) for k in range(3)]
        else:
            neg = self._negatives(epoch.shape[0], epoch_id, 0)
        self.enqueue_epoch(epoch[:, 0], epoch[:, 1], epoch[:, 2], *neg)
        st = self._sync_stats()
        print("[Training Epoch {}], Loss {}".format(epoch_id, st.loss))
        self.writer.add_scalar("model/loss", st.loss_sum, epoch_id)
