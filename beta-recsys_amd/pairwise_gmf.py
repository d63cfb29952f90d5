"""Drop-in ``PairwiseGMF`` / ``PairwiseGMFEngine`` for beta_rec/models/pairwise_gmf.py on libhiprec.so.

The CMN pre-training model ("sibling models on the same kernels").  Interface
parity (file:line = /root/reference/beta_rec/...): ``PairwiseGMF(config)`` models/pairwise_gmf.py:20-66
(``forward(users, items, neg_items) -> (pos_score, neg_score)``, ``predict()`` is a stub there),
``PairwiseGMFEngine(config)`` :69-158 (``train_single_batch(batch) -> float``, ``train_an_epoch``,
its own ``bpr_loss``).  Same FLAT config keys as the reference reads (``n_users n_items emb_dim regs
batch_size lr pretrain_l2_lambda grad_clip neg_count``) next to ``config["model"]`` /
``config["system"]`` which its base class reads; same ``state_dict`` keys and the same initial weights
for the same torch seed.

Forward, loss, backward and the gradient-norm clip run in ``csrc/pgmf.hip``; the optimizer sweep is
the shared ``csrc/optim.hip``.  There is no CPU path.
"""
import ctypes
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .mf import _new_stats, raise_on_status, read_stats
from .ncf import _FlatModel, _ParamView
from .flat_engine import FlatModelEngine
from .torch_engine import HipOptimizer


def truncated_normal_(tensor, mean=0, std=1):
    """models/pairwise_gmf.py:10-17: draw 4 N(0,1) candidates per element, keep the first one inside
    (-2, 2) (candidate 0 when none is), then scale and shift.  Consumes the torch RNG identically."""
    cand = tensor.new_empty(tuple(tensor.shape) + (4,)).normal_()
    inside = (cand < 2) & (cand > -2)
    first = inside.max(-1, keepdim=True)[1]
    tensor.data.copy_(cand.gather(-1, first).squeeze(-1))
    tensor.data.mul_(std).add_(mean)


class PairwiseGMF(_FlatModel):
    """models/pairwise_gmf.py:20-66.  Flat buffer: [user_memory | item_memory | v]."""

    def __init__(self, config):
        super().__init__()
        self.n_users = int(config["n_users"])
        self.n_items = int(config["n_items"])
        self.emb_dim = int(config["emb_dim"])
        v = self._build([("user_memory.weight", (self.n_users, self.emb_dim)),
                         ("item_memory.weight", (self.n_items, self.emb_dim)),
                         ("v.weight", (1, self.emb_dim))])
        # RNG order of pairwise_gmf.py:35-46: nn.Embedding's own N(0,1), truncated normal, twice;
        # nn.Linear's kaiming_uniform_(a=sqrt(5)), then xavier_uniform_
        v["user_memory.weight"].normal_(0, 1)
        truncated_normal_(v["user_memory.weight"], std=0.01)
        v["item_memory.weight"].normal_(0, 1)
        truncated_normal_(v["item_memory.weight"], std=0.01)
        nn.init.kaiming_uniform_(v["v.weight"], a=math.sqrt(5))
        nn.init.xavier_uniform_(v["v.weight"])
        self.user_memory = _ParamView(v["user_memory.weight"])
        self.item_memory = _ParamView(v["item_memory.weight"])
        self.v = _ParamView(v["v.weight"])
        self._stats = None

    def tables(self, flat=None):
        """hiprec_pgmf_tables over the weight buffer (or a same-shaped gradient buffer)."""
        flat = self._flat if flat is None else flat
        base, U, I, D = flat.data_ptr(), self.n_users, self.n_items, self.emb_dim
        return _lib.PgmfTables(base, base + 4 * U * D, base + 4 * (U + I) * D, U, I, D, 0)

    def forward(self, input_users, input_items, input_items_negative):
        """pairwise_gmf.py:48-62 without autograd: ``(pos_score, neg_score)``, each ``[B, 1]``.
        Scoring outside training is not on the hot path; it reuses the bit-exact HIP row gather and
        leaves the D-wide dot to torch on the same device."""
        lib = self._require_hip()
        dev = self._flat.device
        if self._stats is None or self._stats.device != dev:
            self._stats = _new_stats(dev)
        ids = [torch.as_tensor(x, device=dev).to(torch.int64).reshape(-1).contiguous()
               for x in (input_users, input_items, input_items_negative)]
        D = self.emb_dim
        rows = []
        for idx, (name, n_rows) in zip(ids, (("user_memory.weight", self.n_users),
                                             ("item_memory.weight", self.n_items),
                                             ("item_memory.weight", self.n_items))):
            out = torch.empty(idx.numel(), D, dtype=torch.float32, device=dev)
            _lib.check(lib.hiprec_gather_rows(_lib.ptr(self.views()[name]), n_rows, D, _lib.ptr(idx),
                                              idx.numel(), _lib.ptr(out), _lib.ptr(self._stats),
                                              _lib.stream_ptr(dev)))
            rows.append(out)
        st = read_stats(self._stats)
        if st.status:
            self._stats = None
            raise_on_status(st.status)
        w = self.views()["v.weight"]
        return (torch.relu((rows[0] * rows[1]) @ w.t()), torch.relu((rows[0] * rows[2]) @ w.t()))

    def predict(self):
        """pairwise_gmf.py:64-66: a stub in the reference as well."""


class PairwiseGMFEngine(FlatModelEngine):
    """models/pairwise_gmf.py:69-158."""

    def __init__(self, config):
        self.config = config
        self.model = PairwiseGMF(config)
        self.regs = config["regs"]
        self.batch_size = config["batch_size"]
        # pairwise_gmf.py:78 builds Adam(lr=config["lr"]); ModelEngine.__init__ then replaces it when
        # config["model"]["optimizer"] names one of sgd/adam/rmsprop (torch_engine.py:23-39)
        self.optimizer = HipOptimizer("adam", config["lr"])
        super(PairwiseGMFEngine, self).__init__(config)

    def set_optimizer(self):
        name = self.config["model"]["optimizer"] if "optimizer" in self.config["model"] else None
        if name in _lib.OPT_KINDS:
            self.optimizer = HipOptimizer(name, self.config["model"]["lr"])

    def _alloc_extra(self, lib, dev):
        self._ws = torch.zeros(lib.hiprec_pgmf_workspace_bytes(self.model.emb_dim), dtype=torch.uint8, device=dev)
        self._clip_ws = torch.zeros(lib.hiprec_clip_workspace_bytes() // 8, dtype=torch.float64, device=dev)

    def _indices(self, batch_data):
        """pairwise_gmf.py:94-103: ``LongTensor(np.array(x, dtype=np.int32))``; tensors that are
        already on the device skip the host round trip."""
        dev = self.model.flat.device
        out = []
        for x in batch_data:
            if not isinstance(x, torch.Tensor):
                x = torch.from_numpy(np.array(x, dtype=np.int32))
            out.append(x.to(dev).to(torch.int64).reshape(-1).contiguous())
        if not (out[0].numel() == out[1].numel() == out[2].numel()):
            raise ValueError("batch tensors differ in length")
        if out[0].numel() == 0:
            raise ValueError("empty batch")
        return out

    def _enqueue_grad(self, batch_data, clip=True):
        lib = self._setup()
        m = self.model
        dev = m.flat.device
        users, pos, neg = self._indices(batch_data)
        B = users.numel()
        w, g = m.tables(), m.tables(self._g_flat)
        st = _lib.stream_ptr(dev)
        _lib.check(lib.hiprec_pgmf_bpr_grad(
            ctypes.byref(w), ctypes.byref(g), _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), B, self._batch_share() / B,
            # data-parallel replicas: the lambda ||v|| term (loss and gradient) is added once, on rank 0
            float(self.config["pretrain_l2_lambda"]) if self._dp_rank == 0 else 0.0, _lib.ptr(self._stats),
            _lib.ptr(self._scratch),
            self._scratch.numel(), _lib.ptr(self._ws), self._ws.numel(), st))
        if clip:
            _lib.check(lib.hiprec_clip_grad_norm(
                _lib.ptr(self._g_flat), self._g_flat.numel(), float(self.config["grad_clip"]),
                _lib.ptr(self._clip_ws), self._clip_ws.numel() * 8, st))

    def backward_only(self, batch_data, clip=True):
        """zero_grad + forward + loss + backward (+ clip) without the optimizer step:
        ``(loss, grads, total_norm)``; ``total_norm`` is None when ``clip`` is False."""
        self._enqueue_grad(batch_data, clip)
        st, grads = self._finish_backward_only()
        return st.loss, grads, (float(self._clip_ws[0]) if clip else None)

    def train_single_batch(self, batch_data):
        """pairwise_gmf.py:82-116: one step, returns ``batch_loss.item()``."""
        assert hasattr(self, "model"), "Please specify the exact model !"
        self._enqueue_step(batch_data)
        return self._sync_stats().loss

    def enqueue_epoch(self, users, pos, neg):
        """One epoch over resident device arrays in visiting order (batches of ``batch_size``, the last
        one short), enqueued by ``hiprec_pgmf_epoch`` with no host work between steps and no sync."""
        lib = self._setup()
        m, opt = self.model, self.optimizer
        dev = m.flat.device
        users, pos, neg = (x.to(dev).to(torch.int64).reshape(-1).contiguous() for x in (users, pos, neg))
        if not (users.numel() == pos.numel() == neg.numel()):
            raise ValueError("epoch arrays differ in length")
        w, g = m.tables(), m.tables(self._g_flat)
        _lib.check(lib.hiprec_pgmf_epoch(
            ctypes.byref(w), ctypes.byref(g), _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), users.numel(),
            int(self.batch_size), float(self.config["pretrain_l2_lambda"]), float(self.config["grad_clip"]),
            opt.kind, opt.lr, opt.beta1, opt.beta2, opt.eps, _lib.ptr(m.flat), _lib.ptr(self._g_flat),
            _lib.ptr(opt.exp_avg), _lib.ptr(opt.exp_avg_sq), m.flat.numel(), _lib.ptr(self._stats),
            _lib.ptr(self._scratch), self._scratch.numel(), _lib.ptr(self._ws), self._ws.numel(),
            _lib.ptr(self._clip_ws), self._clip_ws.numel() * 8, _lib.stream_ptr(dev)))

    def train_an_epoch(self, train_loader, epoch_id):
        """pairwise_gmf.py:118-142: batches come from ``train_loader.cmn_train_loader(batch_size, False,
        neg_count)`` as ``[B, 3]`` arrays (any iterable of such arrays is accepted too); prints the last
        batch's loss and logs the epoch sum.  The batches are collected, moved to the device in one copy
        and the whole epoch is enqueued from C with one host sync at the end."""
        assert hasattr(self, "model"), "Please specify the exact model !"
        self.model.train()
        self._setup()
        dev = self.model.flat.device
        if hasattr(train_loader, "cmn_train_loader"):
            batches = train_loader.cmn_train_loader(self.batch_size, False, self.config["neg_count"])
        else:
            batches = train_loader
        blocks = [b if isinstance(b, torch.Tensor) else torch.from_numpy(np.array(b, dtype=np.int32)) for b in batches]
        if not blocks:
            raise ValueError("empty epoch")
        if any(b.shape[0] != self.batch_size for b in blocks[:-1]) or blocks[-1].shape[0] > self.batch_size:
            raise ValueError("every batch but the last must hold batch_size triples")
        epoch = torch.cat([b.to(dev).to(torch.int64).reshape(-1, 3) for b in blocks])
        self.enqueue_epoch(epoch[:, 0], epoch[:, 1], epoch[:, 2])
        st = self._sync_stats()
        print("[Training Epoch {}], Loss {}".format(epoch_id, st.loss))
        self.writer.add_scalar("model/loss", st.loss_sum, epoch_id)

    def bpr_loss(self, pos_score, neg_score):
        """pairwise_gmf.py:144-158 on caller-supplied score tensors (utility, not the fused path)."""
        return torch.mean(-1 * torch.log(torch.sigmoid(pos_score - neg_score) + 1e-12))
