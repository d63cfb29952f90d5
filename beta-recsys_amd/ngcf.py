"""Drop-in ``NGCF`` / ``NGCFEngine`` for beta_rec/models/ngcf.py on libhiprec.so.

SURVEY.md §8f rank 4 ("sibling models on the same kernels").  Interface parity (file:line =
/root/reference/beta_rec/...): ``NGCF(config, norm_adj)`` models/ngcf.py:9-100 (``forward(norm_adj) ->
(user, item) embeddings``, ``predict``), ``NGCFEngine(config)`` :103-199 (``train_single_batch(batch)
-> (loss, 0.0)``, ``train_an_epoch``, ``bpr_loss``).  Same config keys (``n_users n_items emb_dim
layer_size mess_dropout regs batch_size norm_adj optimizer lr device_str``), same ``state_dict`` keys in
the same order, same initial weights for the same torch seed.

The sparse propagation is LightGCN's SpMM (``csrc/lightgcn.hip``), the two Linear layers per hop run as
grouped exact-fp32 MFMA GEMMs (``csrc/ncf.hip``), everything else (Hadamard, leaky-ReLU, message dropout,
row normalisation, the loss on the concatenated table and all their backwards) is ``csrc/ngcf.hip``.
Message dropout (training only): ``dropout_rng = "torch_cpu"`` (default) draws each hop's keep mask as
``torch.empty(N, d).bernoulli_(1 - p)`` from the global CPU generator, which replays nn.Dropout's own
draws — the same seed drops the same messages as the reference (pinned by the golden fixtures);
``"device"`` draws the keep bytes on the GPU.  There is no CPU path.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .lightgcn import _csr_from_coo, _slice_rows, build_sliced_graphs
from .mf import _new_stats, raise_on_status, read_stats
from .ncf import _FlatModel, _init_linear_like_torch, _ParamView
from .flat_engine import FlatModelEngine


class NGCF(_FlatModel):
    """models/ngcf.py:9-100.  Flat buffer: [user_embedding | item_embedding | GC_l (w, b) ... | Bi_l (w, b) ...]."""

    def __init__(self, config, norm_adj):
        super().__init__()
        self.config = config
        self.n_users, self.n_items = int(config["n_users"]), int(config["n_items"])
        self.emb_dim = int(config["emb_dim"])
        self.layer_size = list(config["layer_size"])
        self.norm_adj = norm_adj
        self.n_layers = len(self.layer_size)
        if not 1 <= self.n_layers <= _lib.NGCF_MAX_LAYERS:
            raise ValueError(f"NGCF supports 1..{_lib.NGCF_MAX_LAYERS} propagation layers, got {self.n_layers}")
        self.dropout_list = list(config["mess_dropout"])
        self.layer_size = [self.emb_dim] + self.layer_size
        dims = self.layer_size
        spec = [("user_embedding.weight", (self.n_users, self.emb_dim)),
                ("item_embedding.weight", (self.n_items, self.emb_dim))]
        for fam in ("GC_weights", "Bi_weights"):
            for i in range(self.n_layers):
                spec += [(f"{fam}.{i}.weight", (dims[i + 1], dims[i])), (f"{fam}.{i}.bias", (dims[i + 1],))]
        v = self._build(spec)
        # RNG order of ngcf.py:29-46: per hop Linear GC then Linear Bi (reset_parameters), then the two
        # nn.Embedding (N(0,1)), then xavier_uniform_ on both
        for i in range(self.n_layers):
            _init_linear_like_torch(v[f"GC_weights.{i}.weight"], v[f"GC_weights.{i}.bias"])
            _init_linear_like_torch(v[f"Bi_weights.{i}.weight"], v[f"Bi_weights.{i}.bias"])
        v["user_embedding.weight"].normal_(0, 1)
        v["item_embedding.weight"].normal_(0, 1)
        nn.init.xavier_uniform_(v["user_embedding.weight"])
        nn.init.xavier_uniform_(v["item_embedding.weight"])
        # registration order = the reference's state_dict order
        self.dropout = nn.ModuleList(nn.Dropout(p) for p in self.dropout_list[: self.n_layers])
        self.GC_weights = nn.ModuleList(_ParamView(v[f"GC_weights.{i}.weight"], v[f"GC_weights.{i}.bias"])
                                        for i in range(self.n_layers))
        self.Bi_weights = nn.ModuleList(_ParamView(v[f"Bi_weights.{i}.weight"], v[f"Bi_weights.{i}.bias"])
                                        for i in range(self.n_layers))
        self.user_embedding = _ParamView(v["user_embedding.weight"])
        self.item_embedding = _ParamView(v["item_embedding.weight"])
        self.dropout_rng = config["dropout_rng"] if "dropout_rng" in config else "torch_cpu"
        self.dropout_seed = int(config["dropout_seed"]) if "dropout_seed" in config else 0
        self._graph = None
        self._ws = None
        self._stats = None
        self._step = 0

    # ---- graph + workspace ------------------------------------------------------------------------
    def graph(self):
        """CSR and transposed CSR of norm_adj on the parameters' device (built once)."""
        dev = self._flat.device
        if self._graph is not None and self._graph["dev"] == dev:
            return self._graph
        co = self.norm_adj.coalesce().cpu()
        N = self.n_users + self.n_items
        if tuple(co.shape) != (N, N):
            raise ValueError(f"norm_adj is {tuple(co.shape)}, expected ({N}, {N})")
        rows, cols = co.indices()[0], co.indices()[1]
        vals = co.values().to(torch.float32)
        rp, c, v, _ = _csr_from_coo(rows, cols, vals, N, dev)
        rpt, ct, vt, order_t = _csr_from_coo(cols, rows, vals, N, dev)
        nnz = int(vals.numel())
        self._graph = {"dev": dev, "nnz": nnz, "rowptr": rp, "col": c, "val": v,
                       "rowptr_t": rpt, "col_t": ct, "val_t": vt}
        if dev.type == "cuda":
            self._graph["slice_row"] = _slice_rows(rp, c, v, N, nnz)
            self._graph["slice_row_t"] = _slice_rows(rpt, ct, vt, N, nnz)
            # the column-sliced SpMM serves the hops when every hop's input width takes one slice width
            dims = self.layer_size[: self.n_layers]
            lib = _lib.load()
            widths = {int(lib.hiprec_sliced_width(N, d)) for d in dims}
            mode = self.config["spmm"] if "spmm" in self.config else "auto"
            if len(widths) == 1 and 0 not in widths:
                self._graph.update(build_sliced_graphs((rp, c, v), (rpt, ct, vt), order_t, N, max(dims), dev, mode))
        return self._graph

    def workspace(self):
        dev = self._flat.device
        if self._ws is not None and self._ws["dev"] == dev:
            return self._ws
        N, dims = self.n_users + self.n_items, self.layer_size
        dmax = max(dims)
        new = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)  # noqa: E731
        # every buffer an SpMM (atomics on heavy rows) or the loss scatters into sits in ONE region that
        # the library clears with one fill per step: side[l], spmm_tmp[l], d_all
        sizes = [N * dims[i] for i in range(self.n_layers)] * 2 + [N * sum(dims)]
        zero_ws = new(sum(sizes))
        parts = torch.split(zero_ws, sizes)
        ws = {"dev": dev, "zero_ws": zero_ws, "all": new(N, sum(dims)), "d_all": parts[-1].view(N, sum(dims))}
        ws["side"] = [parts[i].view(N, dims[i]) for i in range(self.n_layers)]
        ws["spmm_tmp"] = [parts[self.n_layers + i].view(N, dims[i]) for i in range(self.n_layers)]
        ws["bi_in"] = [new(N, dims[i]) for i in range(self.n_layers)]
        for name in ("sum_pre", "bi_pre", "ego"):
            ws[name] = [new(N, dims[i + 1]) for i in range(self.n_layers)]
        ws["nrm"] = [new(N) for _ in range(self.n_layers)]
        for name in ("d_sum_l", "d_bi_l"):   # per hop: the backward chain of a hop is one launch, all wgrads one group
            ws[name] = [new(N, dims[i + 1]) for i in range(self.n_layers)]
        ws["keep"] = [torch.ones(N, dims[i + 1], dtype=torch.uint8, device=dev) for i in range(self.n_layers)]
        for name in ("d_sum", "d_bi", "d_side", "d_bi_in", "d_ego0", "d_ego1", "sliced_src"):
            ws[name] = new(N, dmax)
        self._ws = ws
        return ws

    def plan(self, g_flat=None, decay=0.0, batch_size=1, keep=None):
        """hiprec_ngcf_plan over the weights (and, for training, a same-shaped gradient buffer).
        keep: per hop a uint8 keep tensor or None."""
        gr, ws = self.graph(), self.workspace()
        N = self.n_users + self.n_items
        p = _lib.NgcfPlan()
        p.a = _lib.Csr(gr["rowptr"].data_ptr(), gr["col"].data_ptr(), gr["val"].data_ptr(), None, N, gr["nnz"],
                       _lib.ptr(gr.get("slice_row")))
        p.at = _lib.Csr(gr["rowptr_t"].data_ptr(), gr["col_t"].data_ptr(), gr["val_t"].data_ptr(), None, N,
                        gr["nnz"], _lib.ptr(gr.get("slice_row_t")))
        p.n_users, p.n_items, p.n_layers = self.n_users, self.n_items, self.n_layers
        for i, d in enumerate(self.layer_size):
            p.dim[i] = d
        p.decay, p.inv_reg_batch = float(decay), 1.0 / float(batch_size)
        at = lambda buf, name: buf.data_ptr() + 4 * self.offset_of(name)  # noqa: E731
        p.e0 = at(self._flat, "user_embedding.weight")
        p.g_e0 = None if g_flat is None else at(g_flat, "user_embedding.weight")
        for i in range(self.n_layers):
            for field, name in (("gc_w", f"GC_weights.{i}.weight"), ("gc_b", f"GC_weights.{i}.bias"),
                                ("bi_w", f"Bi_weights.{i}.weight"), ("bi_b", f"Bi_weights.{i}.bias")):
                getattr(p, field)[i] = at(self._flat, name)
                if g_flat is not None:
                    getattr(p, "g_" + field)[i] = at(g_flat, name)
            for field in ("side", "bi_in", "sum_pre", "bi_pre", "ego", "nrm", "spmm_tmp", "d_sum_l", "d_bi_l"):
                getattr(p, field)[i] = ws[field][i].data_ptr()
            k = None if keep is None else keep[i]
            p.keep[i] = None if k is None else k.data_ptr()
            p.keep_scale[i] = 1.0 if k is None else 1.0 / (1.0 - float(self.dropout_list[i]))
            p.keep_prob[i] = 1.0 - float(self.dropout_list[i])
        p.all, p.d_all = ws["all"].data_ptr(), ws["d_all"].data_ptr()
        for field in ("d_sum", "d_bi", "d_side", "d_bi_in"):
            setattr(p, field, ws[field].data_ptr())
        p.zero_ws, p.zero_ws_floats = ws["zero_ws"].data_ptr(), ws["zero_ws"].numel()
        p.keep_gen = 1 if (keep is not None and self.dropout_rng == "device") else 0
        p.keep_seed, p.keep_step = self.dropout_seed, self._step
        p.d_ego[0], p.d_ego[1] = ws["d_ego0"].data_ptr(), ws["d_ego1"].data_ptr()
        if gr.get("slice_w", 0) > 0:
            p.sa, p.sat, p.slice_w = gr["sliced"][0], gr["sliced_t"][0], gr["slice_w"]
            p.sliced_src, p.sliced_src_floats = ws["sliced_src"].data_ptr(), ws["sliced_src"].numel()
        return p

    def draw_keep_masks(self):
        """Per hop the keep bytes of one training forward (None for a hop without dropout / in eval)."""
        if not self.training:
            return None
        self._require_hip()
        ws = self.workspace()
        N = self.n_users + self.n_items
        self._step += 1
        out = []
        for i in range(self.n_layers):
            p = float(self.dropout_list[i])
            if p == 0.0:                      # at::dropout returns its input untouched: no RNG draw
                out.append(None)
                continue
            buf = ws["keep"][i]
            if self.dropout_rng == "torch_cpu":
                mask = torch.empty(N, self.layer_size[i + 1]).bernoulli_(1 - p)
                buf.copy_(mask.to(torch.uint8), non_blocking=False)
            elif self.dropout_rng == "device":
                pass                          # drawn inside the forward (hiprec_ngcf_plan.keep_gen)
            else:
                raise ValueError(f"unknown dropout_rng {self.dropout_rng!r}: 'torch_cpu' or 'device'")
            out.append(buf)
        return out

    def _device_stats(self):
        dev = self._flat.device
        if self._stats is None or self._stats.device != dev:
            self._stats = _new_stats(dev)
        return self._stats

    # ---- reference API ----------------------------------------------------------------------------
    def forward(self, norm_adj=None):
        """ngcf.py:48-80 without autograd: ``(u_g_embeddings, i_g_embeddings)``, each ``[*, sum dims]``.
        The graph is the one given at construction (the reference always passes that same tensor)."""
        lib = self._require_hip()
        keep = self.draw_keep_masks()
        plan = self.plan(keep=keep)
        _lib.check(lib.hiprec_ngcf_forward(ctypes.byref(plan), 1 if keep is not None else 0,
                                           _lib.stream_ptr(self._flat.device)))
        N, d0 = self.n_users + self.n_items, self.emb_dim
        e0 = self._flat[: N * d0].view(N, d0)      # hop 0's slice of the concatenation is the tables themselves
        return torch.split(torch.cat([e0, self._ws["all"][:, d0:]], dim=1), [self.n_users, self.n_items], dim=0)

    def predict(self, users, items):
        """ngcf.py:82-100: a full forward (dropout follows ``self.training`` exactly like the reference,
        which never switches modes here) and the dot product of the concatenated rows."""
        lib = self._require_hip()
        dev = self._flat.device
        users_t, items_t = (x.to(dev, torch.int64).reshape(-1).contiguous() if torch.is_tensor(x) else
                            torch.as_tensor(np.asarray(x), dtype=torch.int64).to(dev).reshape(-1).contiguous()
                            for x in (users, items))
        if users_t.numel() != items_t.numel():
            raise ValueError("users and items differ in length")
        stats = self._device_stats()
        keep = self.draw_keep_masks()
        plan = self.plan(keep=keep)
        st = _lib.stream_ptr(dev)
        _lib.check(lib.hiprec_ngcf_forward(ctypes.byref(plan), 1 if keep is not None else 0, st))
        scores = torch.empty(users_t.numel(), dtype=torch.float32, device=dev)
        _lib.check(lib.hiprec_ngcf_predict(ctypes.byref(plan), _lib.ptr(users_t), _lib.ptr(items_t),
                                           users_t.numel(), _lib.ptr(scores), _lib.ptr(stats), st))
        s = read_stats(stats)
        if s.status:
            self._stats = None
            raise_on_status(s.status)
        return scores


class NGCFEngine(FlatModelEngine):
    """models/ngcf.py:103-199."""

    def __init__(self, config):
        self.config = config
        self.regs = config["model"]["regs"]
        self.decay = self.regs[0]
        self.batch_size = config["model"]["batch_size"]
        self.norm_adj = config["model"]["norm_adj"]
        self.model = NGCF(config["model"], self.norm_adj)
        super(NGCFEngine, self).__init__(config)
        self.model.to(self.device)

    def _enqueue_grad(self, batch_data):
        lib = self._setup()
        m = self.model
        dev = m.flat.device
        users, pos, neg = (torch.as_tensor(x, device=dev).to(torch.int64).reshape(-1).contiguous()
                           for x in batch_data)
        B = users.numel()
        if not (pos.numel() == B and neg.numel() == B):
            raise ValueError("batch tensors differ in length")
        if B == 0:
            raise ValueError("empty batch")
        plan = m.plan(self._g_flat, self.decay, self.batch_size, m.draw_keep_masks())
        _lib.check(lib.hiprec_ngcf_grad(ctypes.byref(plan), _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), B,
                                        self._batch_share() / B, _lib.ptr(self._stats), _lib.ptr(self._scratch),
                                        self._scratch.numel(), _lib.stream_ptr(dev)))

    def train_single_batch(self, batch_data):
        """ngcf.py:118-149: one step, returns ``(batch_loss.item(), batch_reg_loss)`` — the second is the
        constant 0.0 of ngcf.py:198."""
        assert hasattr(self, "model"), "Please specify the exact model !"
        self._enqueue_step(batch_data)
        return self._sync_stats().loss, 0.0

    def train_an_epoch(self, train_loader, epoch_id):
        """ngcf.py:151-170: prints the last batch's loss and the (zero) regulariser sum, logs both."""
        assert hasattr(self, "model"), "Please specify the exact model !"
        self.model.train()
        lib = self._setup()
        _lib.check(lib.hiprec_stats_begin_epoch(_lib.ptr(self._stats),
                                                _lib.stream_ptr(self.model.flat.device)))
        for batch_data in train_loader:
            self._enqueue_step(batch_data)
        st = self._sync_stats()
        regularizer = 0.0
        print(f"[Training Epoch {epoch_id}], Loss {st.loss}, Regularizer {regularizer}")
        self.writer.add_scalar("model/loss", st.loss_sum, epoch_id)
        self.writer.add_scalar("model/regularizer", regularizer, epoch_id)

    def bpr_loss(self, users, pos_items, neg_items):
        """ngcf.py:172-199 on caller-supplied embedding rows (utility, not the fused path)."""
        pos_scores = torch.sum(torch.mul(users, pos_items), dim=1)
        neg_scores = torch.sum(torch.mul(users, neg_items), dim=1)
        regularizer = (0.5 * (users ** 2).sum() + 0.5 * (pos_items ** 2).sum() + 0.5 * (neg_items ** 2).sum())
        regularizer = regularizer / self.batch_size
        mf_loss = -torch.mean(torch.nn.functional.logsigmoid(pos_scores - neg_scores))
        return mf_loss, self.decay * regularizer, 0.0
