"""Drop-in NCF family for beta_rec/models/{ncf,gmf,mlp}.py backed by libhiprec.so (HIP, gfx950).

Interface parity (file:line = /root/reference/beta_rec/...):

* ``NeuMF`` / ``NeuMFEngine``   models/ncf.py:12-193  (incl. ``init_weights`` quirk Q8 and the
  ``ncf_pre`` merge of GMF + MLP checkpoints, models/ncf.py:155-193)
* ``GMF`` / ``GMFEngine``       models/gmf.py:8-100
* ``MLP`` / ``MLPEngine``       models/mlp.py:8-116

Same config keys, same ``state_dict`` keys / shapes (``embedding_*``, ``fc_layers.{1,4,7}.*``,
``affine_output.*``), same initial weights for the same torch seed, ``train_single_batch(users,
items, ratings) -> float``, ``train_an_epoch`` with the reference's prints and ``add_scalar`` tag,
``model.predict -> [n, 1]`` tensor.  The arithmetic — gather, fp32-MFMA tower forward / backward,
BCE, scatter, dense optimizer — is ``csrc/ncf.hip`` + ``csrc/optim.hip``.  Tower dropout (``dropout`` > 0,
models/ncf.py:42-45; the shipped configs use 0) is supported: the keep masks are either ``nn.Dropout``'s own
CPU draws replayed (same torch seed -> the reference's masks) or drawn on the device (``dropout_rng``).
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn
from torch.nn import Parameter

from . import _lib
from .mf import _new_stats, clear_status, raise_on_status, read_stats, timeit
from .torch_engine import ModelEngine


class _ParamView(nn.Module):
    """A module whose parameters (``weight`` and optionally ``bias``) are views of a flat buffer."""

    def __init__(self, weight, bias=None):
        super().__init__()
        self.weight = Parameter(weight, requires_grad=False)
        if bias is not None:
            self.bias = Parameter(bias, requires_grad=False)

    def extra_repr(self):
        return "x".join(str(s) for s in self.weight.shape)


class _FlatModel(nn.Module):
    """Base: named parameter views over ONE flat fp32 buffer (tables first, dense layers after)."""

    def _build(self, spec):
        """spec: list of (name, shape); allocates the flat buffer and returns the views by name."""
        self._spec = [(n, tuple(s)) for n, s in spec]
        sizes = [int(np.prod(s)) for _, s in self._spec]
        self._offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self._flat = torch.zeros(int(self._offsets[-1]), dtype=torch.float32)
        return self.views()

    def views(self, flat=None):
        flat = self._flat if flat is None else flat
        return {n: flat[self._offsets[k]:self._offsets[k + 1]].view(*s)
                for k, (n, s) in enumerate(self._spec)}

    def offset_of(self, name):
        return int(self._offsets[[n for n, _ in self._spec].index(name)])

    def _owner(self, name):
        mod = self
        parts = name.split(".")
        for p in parts[:-1]:
            mod = getattr(mod, p) if not p.isdigit() else mod[int(p)]
        return mod, parts[-1]

    def _rebind(self, flat):
        self._flat = flat
        for name, view in self.views(flat).items():
            mod, attr = self._owner(name)
            getattr(mod, attr).data = view

    def _apply(self, fn, recurse=True):
        new_flat = fn(self._flat)
        if new_flat.dtype != torch.float32:
            raise TypeError("hiprec models keep fp32 parameters (the reference trains in fp32)")
        if new_flat is not self._flat:
            self._rebind(new_flat.contiguous())
        return self

    @property
    def flat(self):
        return self._flat

    def _require_hip(self):
        if self._flat.device.type != "cuda":
            raise RuntimeError(
                "hiprec NCF models compute on an MI355X through libhiprec.so only; parameters are "
                f"on {self._flat.device} and there is deliberately no CPU fallback")
        return _lib.load()


def _tower_dims(emb_dim, n_layers):
    """models/ncf.py:40-46: layer i maps emb_dim*2^(L-i) -> half."""
    return [(emb_dim * 2 ** (n_layers - i), emb_dim * 2 ** (n_layers - i) // 2) for i in range(n_layers)]


def _make_fc_layers(views, dims, dropout):
    mods = []
    for i, _ in enumerate(dims):
        mods += [nn.Dropout(p=dropout),
                 _ParamView(views[f"fc_layers.{3 * i + 1}.weight"], views[f"fc_layers.{3 * i + 1}.bias"]),
                 nn.ReLU()]
    return nn.Sequential(*mods)


class _NcfBase(_FlatModel):
    """Shared forward / predict plumbing of NeuMF, GMF and MLP."""

    kind = None

    def _finish(self, config):
        self.config = config
        self._plan_cache = None
        self._ws = None
        self._stats = None

    # -- plan / workspace ---------------------------------------------------------------------
    def _names(self):
        if self.kind == "neumf":
            return ("embedding_user_mlp.weight", "embedding_item_mlp.weight",
                    "embedding_user_mf.weight", "embedding_item_mf.weight")
        if self.kind == "gmf":
            return (None, None, "embedding_user.weight", "embedding_item.weight")
        return ("embedding_user.weight", "embedding_item.weight", None, None)

    def table_floats(self):
        """How many leading floats of the flat buffers are embedding tables (they come first in the layout)."""
        tables = {n for n in self._names() if n is not None}
        end = 0
        for k, (name, _) in enumerate(self._spec):
            if name in tables:
                end = max(end, int(self._offsets[k + 1]))
        first_dense = min((int(self._offsets[k]) for k, (name, _) in enumerate(self._spec) if name not in tables),
                          default=end)
        return end if end <= first_dense else 0

    def workspace(self, batch):
        """Activation buffers for up to ``batch`` samples (grown on demand)."""
        dev = self._flat.device
        if self._ws is not None and self._ws["max_batch"] >= batch and self._ws["dev"] == dev:
            return self._ws
        dims = self.tower_dims
        ws = {"max_batch": int(batch), "dev": dev, "act": [], "dact": []}
        if self.dim_mlp > 0:
            widths = [2 * self.dim_mlp] + [o for _, o in dims]
            ws["act"] = [torch.empty(batch, w, device=dev) for w in widths]
            ws["dact"] = [torch.empty(batch, w, device=dev) for w in widths]
            # dropout keep bytes of every Linear's input (only touched when dropout > 0 in training)
            ws["keep"] = [torch.ones(batch, nin, dtype=torch.uint8, device=dev) for nin, _ in dims]
        ws["mf"] = torch.empty(batch, max(self.dim_mf, 1), device=dev)
        ws["dmf"] = torch.empty(batch, max(self.dim_mf, 1), device=dev)
        ws["scores"] = torch.empty(batch, device=dev)
        self._ws = ws
        self._plan_cache = None
        return ws

    def plan(self, batch, g_flat=None):
        """hiprec_ncf_plan over the parameters (and the gradient buffer ``g_flat`` if given)."""
        ws = self.workspace(batch)
        key = (self._flat.data_ptr(), None if g_flat is None else g_flat.data_ptr(), ws["max_batch"])
        if self._plan_cache is None:
            self._plan_cache = {}
        if key in self._plan_cache:   # the forward (eval) plan and the gradient plan coexist
            return self._plan_cache[key]
        w = self.views()
        g = self.views(g_flat) if g_flat is not None else None
        p = _lib.NcfPlan()
        um, im, ug, ig = self._names()
        for field, name in (("user_mlp", um), ("item_mlp", im), ("user_mf", ug), ("item_mf", ig)):
            if name is not None:
                setattr(p, field, w[name].data_ptr())
                if g is not None:
                    setattr(p, "g_" + field, g[name].data_ptr())
        p.n_users, p.n_items = self.n_users, self.n_items
        p.dim_mlp, p.dim_mf = self.dim_mlp, self.dim_mf
        p.n_layers = len(self.tower_dims) if self.dim_mlp > 0 else 0
        p.relu_input = 1 if self.kind == "neumf" else 0
        for l, (nin, nout) in enumerate(self.tower_dims if self.dim_mlp > 0 else []):
            p.layer_in[l], p.layer_out[l] = nin, nout
            p.fc_w[l] = w[f"fc_layers.{3 * l + 1}.weight"].data_ptr()
            p.fc_b[l] = w[f"fc_layers.{3 * l + 1}.bias"].data_ptr()
            if g is not None:
                p.g_fc_w[l] = g[f"fc_layers.{3 * l + 1}.weight"].data_ptr()
                p.g_fc_b[l] = g[f"fc_layers.{3 * l + 1}.bias"].data_ptr()
        p.out_w, p.out_b = w["affine_output.weight"].data_ptr(), w["affine_output.bias"].data_ptr()
        if g is not None:
            p.g_out_w = g["affine_output.weight"].data_ptr()
            p.g_out_b = g["affine_output.bias"].data_ptr()
        p.max_batch = ws["max_batch"]
        for l, t in enumerate(ws["act"]):
            p.act[l] = t.data_ptr()
            p.dact[l] = ws["dact"][l].data_ptr()
        p.mf, p.dmf, p.scores = ws["mf"].data_ptr(), ws["dmf"].data_ptr(), ws["scores"].data_ptr()
        p.keep_scale = 1.0
        if len(self._plan_cache) > 4:
            self._plan_cache.clear()
        self._plan_cache[key] = p
        return p

    def draw_keep_masks(self, plan, batch):
        """Training step with dropout > 0: draw the keep bytes of every Linear's input (ncf.py:42-45,
        mlp.py:30-33) and point ``plan`` at them; otherwise clear the plan's keep pointers.
        ``dropout_rng = "torch_cpu"`` (default) replays nn.Dropout's own CPU draws — one
        ``torch.empty(B, in_features).bernoulli_(1 - p)`` per layer, in layer order, so the same torch seed
        drops the same activations as the reference (pinned by tests/golden/ncf_*_dropout.npz);
        ``"device"`` draws them on the GPU."""
        p = float(getattr(self, "dropout", 0.0) or 0.0)
        n_layers = len(self.tower_dims) if self.dim_mlp > 0 else 0
        active = self.training and p > 0.0 and n_layers > 0
        for l in range(n_layers):
            plan.keep[l] = None
        plan.keep_scale = 1.0
        if not active:
            return
        ws = self.workspace(batch)
        rng = self.config["dropout_rng"] if "dropout_rng" in self.config else "torch_cpu"
        self._dropout_step = getattr(self, "_dropout_step", 0) + 1
        for l, (nin, _) in enumerate(self.tower_dims):
            buf = ws["keep"][l]
            if rng == "torch_cpu":
                buf[:batch].copy_(torch.empty(batch, nin).bernoulli_(max(1 - p, 0.0)).to(torch.uint8))
            elif rng == "device":
                seed = int(self.config["dropout_seed"]) if "dropout_seed" in self.config else 0
                _lib.check(_lib.load().hiprec_edge_dropout_mask(
                    _lib.ptr(buf), batch * nin, 1.0 - p, seed * 64 + l, self._dropout_step,
                    _lib.stream_ptr(self._flat.device)))
            else:
                raise ValueError(f"unknown dropout_rng {rng!r}: 'torch_cpu' or 'device'")
            plan.keep[l] = buf.data_ptr()
        plan.keep_scale = 1.0 / (1.0 - p) if p < 1.0 else 0.0   # nn.Dropout(p=1) outputs zeros

    # -- reference API ------------------------------------------------------------------------
    def forward(self, user_indices, item_indices):
        """models/ncf.py:52-71 / gmf.py:29-36 / mlp.py:40-51 without autograd: [n, 1] ratings."""
        lib = self._require_hip()
        dev = self._flat.device
        users = torch.as_tensor(user_indices, device=dev).to(torch.int64).reshape(-1).contiguous()
        items = torch.as_tensor(item_indices, device=dev).to(torch.int64).reshape(-1).contiguous()
        if users.numel() != items.numel():
            raise ValueError("user_indices and item_indices must have the same length")
        n = users.numel()
        if self._stats is None or self._stats.device != dev:
            self._stats = _new_stats(dev)
        plan = self.plan(max(n, 1))
        if n:
            self.draw_keep_masks(plan, n)   # like the reference, dropout follows self.training here too
        _lib.check(lib.hiprec_ncf_forward(ctypes.byref(plan), _lib.ptr(users), _lib.ptr(items), n,
                                          _lib.ptr(self._stats), _lib.stream_ptr(dev)))
        return self._ws["scores"][:n].clone().view(n, 1)

    def predict(self, user_indices, item_indices):
        """models/ncf.py:73-78: numpy / list ids in, [n, 1] score tensor on the device out."""
        dev = self._flat.device
        u = torch.as_tensor(np.asarray(user_indices), dtype=torch.int64).to(dev)
        i = torch.as_tensor(np.asarray(item_indices), dtype=torch.int64).to(dev)
        scores = self.forward(u, i)
        st = read_stats(self._stats)
        if st.status:
            self._stats = None
            raise_on_status(st.status)
        return scores

    def init_weight(self):
        """models/ncf.py:80-82 (a no-op there); GMF / MLP override."""


class NeuMF(_NcfBase):
    """models/ncf.py:12-82."""

    kind = "neumf"

    def __init__(self, config):
        super().__init__()
        self.n_users, self.n_items = int(config["n_users"]), int(config["n_items"])
        self.emb_dim = int(config["emb_dim"])
        self.n_layers = int(config["mlp_config"]["n_layers"])
        self.dropout = config["dropout"]
        self.latent_dim_mlp = self.emb_dim * (2 ** self.n_layers) // 2
        self.latent_dim_gmf = self.emb_dim
        self.dim_mlp, self.dim_mf = self.latent_dim_mlp, self.latent_dim_gmf
        self.tower_dims = _tower_dims(self.emb_dim, self.n_layers)
        U, I, Dm, E = self.n_users, self.n_items, self.dim_mlp, self.dim_mf
        spec = [("embedding_user_mlp.weight", (U, Dm)), ("embedding_item_mlp.weight", (I, Dm)),
                ("embedding_user_mf.weight", (U, E)), ("embedding_item_mf.weight", (I, E))]
        for i, (nin, nout) in enumerate(self.tower_dims):
            spec += [(f"fc_layers.{3 * i + 1}.weight", (nout, nin)), (f"fc_layers.{3 * i + 1}.bias", (nout,))]
        spec += [("affine_output.weight", (1, 2 * E)), ("affine_output.bias", (1,))]
        v = self._build(spec)
        # consume the torch RNG in the order of models/ncf.py:27-49: four nn.Embedding, the Linear
        # layers of the tower, affine_output
        for name in ("embedding_user_mlp.weight", "embedding_item_mlp.weight",
                     "embedding_user_mf.weight", "embedding_item_mf.weight"):
            v[name].normal_(0, 1)
        for i, (nin, nout) in enumerate(self.tower_dims):
            _init_linear_like_torch(v[f"fc_layers.{3 * i + 1}.weight"], v[f"fc_layers.{3 * i + 1}.bias"])
        _init_linear_like_torch(v["affine_output.weight"], v["affine_output.bias"])
        self.embedding_user_mlp = _ParamView(v["embedding_user_mlp.weight"])
        self.embedding_item_mlp = _ParamView(v["embedding_item_mlp.weight"])
        self.embedding_user_mf = _ParamView(v["embedding_user_mf.weight"])
        self.embedding_item_mf = _ParamView(v["embedding_item_mf.weight"])
        self.fc_layers = _make_fc_layers(v, self.tower_dims, self.dropout)
        self.affine_output = _ParamView(v["affine_output.weight"], v["affine_output.bias"])
        self.logistic = nn.Sigmoid()
        self._finish(config)


class GMF(_NcfBase):
    """models/gmf.py:8-48."""

    kind = "gmf"

    def __init__(self, config):
        super().__init__()
        self.num_users = self.n_users = int(config["n_users"])
        self.num_items = self.n_items = int(config["n_items"])
        self.emb_dim = int(config["emb_dim"])
        self.dim_mlp, self.dim_mf, self.tower_dims = 0, self.emb_dim, []
        v = self._build([("embedding_user.weight", (self.n_users, self.emb_dim)),
                         ("embedding_item.weight", (self.n_items, self.emb_dim)),
                         ("affine_output.weight", (1, self.emb_dim)), ("affine_output.bias", (1,))])
        v["embedding_user.weight"].normal_(0, 1)
        v["embedding_item.weight"].normal_(0, 1)
        self.embedding_user = _ParamView(v["embedding_user.weight"])
        self.embedding_item = _ParamView(v["embedding_item.weight"])
        self.init_weight()
        _init_linear_like_torch(v["affine_output.weight"], v["affine_output.bias"])
        self.affine_output = _ParamView(v["affine_output.weight"], v["affine_output.bias"])
        self.logistic = nn.Sigmoid()
        self._finish(config)

    def init_weight(self):
        """models/gmf.py:45-48 — quirk Q8: the USER table twice, the item table stays N(0,1)."""
        nn.init.normal_(self.embedding_user.weight, std=0.01)
        nn.init.normal_(self.embedding_user.weight, std=0.01)


class MLP(_NcfBase):
    """models/mlp.py:8-63."""

    kind = "mlp"

    def __init__(self, config):
        super().__init__()
        self.n_users, self.n_items = int(config["n_users"]), int(config["n_items"])
        self.emb_dim = int(config["emb_dim"])
        self.n_layers = int(config["mlp_config"]["n_layers"])
        self.dropout = config["dropout"]
        self.latent_dim = self.emb_dim * (2 ** self.n_layers) // 2
        self.dim_mlp, self.dim_mf = self.latent_dim, 0
        self.tower_dims = _tower_dims(self.emb_dim, self.n_layers)
        spec = [("embedding_user.weight", (self.n_users, self.latent_dim)),
                ("embedding_item.weight", (self.n_items, self.latent_dim))]
        for i, (nin, nout) in enumerate(self.tower_dims):
            spec += [(f"fc_layers.{3 * i + 1}.weight", (nout, nin)), (f"fc_layers.{3 * i + 1}.bias", (nout,))]
        spec += [("affine_output.weight", (1, self.emb_dim)), ("affine_output.bias", (1,))]
        v = self._build(spec)
        v["embedding_user.weight"].normal_(0, 1)
        v["embedding_item.weight"].normal_(0, 1)
        self.embedding_user = _ParamView(v["embedding_user.weight"])
        self.embedding_item = _ParamView(v["embedding_item.weight"])
        self.init_weight()
        for i, _ in enumerate(self.tower_dims):
            _init_linear_like_torch(v[f"fc_layers.{3 * i + 1}.weight"], v[f"fc_layers.{3 * i + 1}.bias"])
        _init_linear_like_torch(v["affine_output.weight"], v["affine_output.bias"])
        self.fc_layers = _make_fc_layers(v, self.tower_dims, self.dropout)
        self.affine_output = _ParamView(v["affine_output.weight"], v["affine_output.bias"])
        self.logistic = nn.Sigmoid()
        self._finish(config)

    def init_weight(self):
        """models/mlp.py:60-63 — same quirk as GMF."""
        nn.init.normal_(self.embedding_user.weight, std=0.01)
        nn.init.normal_(self.embedding_user.weight, std=0.01)


def _init_linear_like_torch(weight, bias):
    """nn.Linear.reset_parameters on views: kaiming_uniform_(a=sqrt(5)) then the bias bound."""
    nn.init.kaiming_uniform_(weight, a=5 ** 0.5)
    fan_in = weight.shape[1]
    bound = 1 / fan_in ** 0.5 if fan_in > 0 else 0
    nn.init.uniform_(bias, -bound, bound)


class _NcfEngine(ModelEngine):
    """Shared engine: BCE step through hiprec_ncf_grad + one dense optimizer sweep."""

    print_last_loss = False  # NeuMFEngine prints the last batch's loss, GMF/MLP the epoch sum

    def _post_init(self):
        self.model.to(self.device)
        self._ready = False

    def _setup(self):
        lib = self.require_hip()
        flat = self.model.flat
        if self._ready and self._g_flat.device == flat.device:
            return lib
        self._g_flat = torch.zeros_like(flat)
        self.optimizer.allocate_state(flat)
        self._scratch = torch.zeros(lib.hiprec_scratch_bytes(0), dtype=torch.uint8, device=flat.device)
        self._stats = _new_stats(flat.device, self.optimizer.beta1 or 0.9, self.optimizer.beta2 or 0.999)
        self._ready = True
        return lib

    def _enqueue_step(self, users, items, ratings):
        lib = self._setup()
        m, opt = self.model, self.optimizer
        dev = m.flat.device
        users = torch.as_tensor(users, device=dev).to(torch.int64).reshape(-1).contiguous()
        items = torch.as_tensor(items, device=dev).to(torch.int64).reshape(-1).contiguous()
        ratings = torch.as_tensor(ratings, device=dev).to(torch.float32).reshape(-1).contiguous()
        B = users.numel()
        if not (items.numel() == B and ratings.numel() == B):
            raise ValueError("users, items and ratings differ in length")
        if B == 0:
            raise ValueError("empty batch")
        st = _lib.stream_ptr(dev)
        plan = m.plan(B, self._g_flat)
        m.draw_keep_masks(plan, B)
        # forward + backward + optimizer.step() in one call: the tables' share of the dense sweep rides in the grouped
        # weight-gradient launch (csrc/ncf.hip hiprec_ncf_step)
        _lib.check(lib.hiprec_ncf_step(
            ctypes.byref(plan), _lib.ptr(users), _lib.ptr(items), _lib.ptr(ratings), B, 1.0 / B, opt.kind,
            _lib.ptr(m.flat), _lib.ptr(self._g_flat), _lib.ptr(opt.exp_avg), _lib.ptr(opt.exp_avg_sq), m.flat.numel(),
            m.table_floats(), m.offset_of("affine_output.bias"), opt.lr, opt.beta1, opt.beta2, opt.eps,
            _lib.ptr(self._stats), _lib.ptr(self._scratch), self._scratch.numel(), st))

    def _sync_stats(self):
        st = read_stats(self._stats)
        if st.status:
            clear_status(self._stats)
            raise_on_status(st.status)
        return st

    def backward_only(self, users, items, ratings):
        """zero_grad + forward + backward without the optimizer step: ``(loss, grads dict)``."""
        lib = self._setup()
        m = self.model
        dev = m.flat.device
        users = torch.as_tensor(users, device=dev).to(torch.int64).contiguous()
        items = torch.as_tensor(items, device=dev).to(torch.int64).contiguous()
        ratings = torch.as_tensor(ratings, device=dev).to(torch.float32).contiguous()
        B = users.numel()
        st = _lib.stream_ptr(dev)
        plan = m.plan(B, self._g_flat)
        m.draw_keep_masks(plan, B)
        _lib.check(lib.hiprec_ncf_grad(
            ctypes.byref(plan), _lib.ptr(users), _lib.ptr(items), _lib.ptr(ratings), B, 1.0 / B,
            _lib.ptr(self._stats), _lib.ptr(self._scratch), self._scratch.numel(), st))
        g = m.views(self._g_flat)
        _lib.check(lib.hiprec_finalize_stats(_lib.ptr(self._stats), _lib.ptr(self._scratch),
                                             _lib.ptr(g["affine_output.bias"]), None, st))
        stt = self._sync_stats()
        grads = {k: v.clone() for k, v in g.items()}
        self._g_flat.zero_()
        return stt.loss, grads

    def load_optimizer_state(self, step, exp_avg=None, exp_avg_sq=None):
        """Restore the optimizer clock and moments (dicts keyed like ``state_dict``)."""
        lib = self._setup()
        opt, m = self.optimizer, self.model
        dev = m.flat.device
        _lib.check(lib.hiprec_stats_reset(_lib.ptr(self._stats), opt.beta1 or 0.9, opt.beta2 or 0.999,
                                          _lib.stream_ptr(dev)))
        _lib.check(lib.hiprec_stats_set_step(_lib.ptr(self._stats), int(step), opt.beta1 or 0.9,
                                             opt.beta2 or 0.999, _lib.stream_ptr(dev)))
        for buf, src in ((opt.exp_avg, exp_avg), (opt.exp_avg_sq, exp_avg_sq)):
            if buf is None:
                continue
            if src is None:
                buf.zero_()
                continue
            for name, view in m.views(buf).items():
                view.copy_(torch.as_tensor(src[name], dtype=torch.float32).reshape(view.shape))

    def train_single_batch(self, users, items, ratings):
        """models/ncf.py:100-120: one optimisation step, returns the batch loss as a float."""
        assert hasattr(self, "model"), "Please specify the exact model !"
        self._enqueue_step(users, items, ratings)
        return self._sync_stats().loss

    @timeit
    def train_an_epoch(self, train_loader, epoch_id):
        """models/ncf.py:122-140 (gmf.py:82-100, mlp.py:98-116): one host sync per epoch."""
        assert hasattr(self, "model"), "Please specify the exact model !"
        self.model.train()
        lib = self._setup()
        _lib.check(lib.hiprec_stats_begin_epoch(_lib.ptr(self._stats),
                                                _lib.stream_ptr(self.model.flat.device)))
        for batch_id, batch in enumerate(train_loader):
            user, item, rating = batch[0], batch[1], batch[2]
            self._enqueue_step(user, item, torch.as_tensor(rating).float())
        st = self._sync_stats()
        shown = st.loss if self.print_last_loss else st.loss_sum
        print("[Training Epoch {}], Loss {}".format(epoch_id, shown))
        self.writer.add_scalar("model/loss", st.loss_sum, epoch_id)


class NeuMFEngine(_NcfEngine):
    """models/ncf.py:85-193."""

    print_last_loss = True  # models/ncf.py:139 prints `loss` (the last batch), quirk Q5 family

    def __init__(self, config):
        self.config = config
        self.model = NeuMF(config["model"])
        self.loss = torch.nn.BCELoss()
        super(NeuMFEngine, self).__init__(config)
        print(self.model)
        if self.config["model"]["model"] == "ncf_pre":
            self.load_pretrain_weights()
        else:
            self.init_weights()
        self._post_init()

    def init_weights(self):
        """models/ncf.py:142-153, quirk Q8 included: ``embedding_user_mlp`` is initialised twice and
        ``embedding_item_mlp`` never (it keeps nn.Embedding's N(0,1))."""
        m = self.model
        on_cpu = m.flat.device.type == "cpu"
        if not on_cpu:  # draw on the CPU generator like the reference, then move back
            dev = m.flat.device
            m.to("cpu")
        nn.init.normal_(m.embedding_user_mf.weight, std=0.01)
        nn.init.normal_(m.embedding_item_mf.weight, std=0.01)
        nn.init.normal_(m.embedding_user_mlp.weight, std=0.01)
        nn.init.normal_(m.embedding_user_mlp.weight, std=0.01)
        for m1 in m.fc_layers:
            if isinstance(m1, _ParamView):
                nn.init.xavier_uniform_(m1.weight)
        nn.init.kaiming_uniform_(m.affine_output.weight, a=1, nonlinearity="sigmoid")
        if not on_cpu:
            m.to(dev)

    def load_pretrain_weights(self):
        """models/ncf.py:155-193: merge trained GMF and MLP checkpoints into NeuMF."""
        gmf_model = GMF(self.config["model"])
        gmf_save_dir = os.path.join(self.config["system"]["model_save_dir"],
                                    self.config["model"]["gmf_config"]["save_name"])
        self.resume_checkpoint(gmf_save_dir, gmf_model)
        mlp_model = MLP(self.config["model"])
        mlp_save_dir = os.path.join(self.config["system"]["model_save_dir"],
                                    self.config["model"]["mlp_config"]["save_name"])
        self.resume_checkpoint(mlp_save_dir, mlp_model)
        m = self.model
        with torch.no_grad():
            m.embedding_user_mf.weight.copy_(gmf_model.embedding_user.weight)
            m.embedding_item_mf.weight.copy_(gmf_model.embedding_item.weight)
            m.embedding_user_mlp.weight.copy_(mlp_model.embedding_user.weight)
            m.embedding_item_mlp.weight.copy_(mlp_model.embedding_item.weight)
            for m1, m2 in zip(m.fc_layers, mlp_model.fc_layers):
                if isinstance(m1, _ParamView) and isinstance(m2, _ParamView):
                    m1.weight.copy_(m2.weight)
                    m1.bias.copy_(m2.bias)
            m.affine_output.weight.copy_(0.5 * torch.cat(
                [mlp_model.affine_output.weight, gmf_model.affine_output.weight], dim=-1))
            m.affine_output.bias.copy_(0.5 * (mlp_model.affine_output.bias + gmf_model.affine_output.bias))


class GMFEngine(_NcfEngine):
    """models/gmf.py:51-100."""

    def __init__(self, config):
        self.model = GMF(config["model"])
        self.loss = torch.nn.BCELoss()
        super(GMFEngine, self).__init__(config)
        self._post_init()


class MLPEngine(_NcfEngine):
    """models/mlp.py:66-116."""

    def __init__(self, config):
        self.model = MLP(config["model"])
        self.loss = torch.nn.BCELoss()
        super(MLPEngine, self).__init__(config)
        self._post_init()
