"""Row-sharded NCF family across the GPUs of one node ("NCF: same table sharding; tower weights
replicated + gradient all-reduce").

The reference trains NeuMF / GMF / MLP on one device (beta_rec/models/ncf.py:100-120, gmf.py, mlp.py).  Here the
embedding tables -- the four of NeuMF (``embedding_{user,item}_{mlp,mf}``, models/ncf.py:27-38), the two of GMF / MLP
-- are row-sharded exactly like the MF tables of sharded.py (``owner(row) = row mod R``, local index ``row // R``),
while the tower (``fc_layers``) and the head (``affine_output``) are replicated: classic data parallelism for the
dense part.  One process per GPU, ``torch.distributed`` over RCCL/xGMI.

One step on a global batch (every rank contributes its ``b_r`` samples; 1/B uses the global B, so the result equals
the single-process reference on the concatenated batch up to fp32 summation order):

    A2A   user ids -> owner(user), rows of the user tables back      (dim_mlp + dim_mf floats / sample)
    A2A   item ids -> owner(item), rows of the item tables back
    step  the single-GPU NCF kernels (csrc/ncf.hip) on (fetched rows, replicated tower): forward, BCE, backward
    A2A   row gradients -> their owners, scatter-add into the local dense gradient   (x 2 sides)
    all-reduce of [tower + head gradient | loss]   (~170 kB at the ncf_default.json shape)
    local dense optimizer sweep over [table shards | tower | head]

The fetched buffers simply play the four tables for ``hiprec_ncf_grad`` (sample k reads row k), so the kernels are
the single-GPU ones.  The CPU tests drive this host code over gloo with the numpy oracle standing in for the kernels
through the ``kernels=`` seam; the product default is :class:`HipNcfKernels`, which refuses to run without a GPU.
"""
import contextlib
import ctypes
import io

import torch
from . import _dist as dist

from . import _lib
from . import ncf as ncf_mod
from .mf import _new_stats, clear_status, raise_on_status, read_stats
from .sharded import shard_rows
from .torch_engine import HipOptimizer


class HipNcfKernels:
    """The product compute backend: libhiprec through the C ABI (no fallback)."""

    def __init__(self, device):
        if device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("the sharded NCF engine computes through libhiprec.so on an MI355X; there is "
                               f"deliberately no CPU fallback (device {device})")
        self.lib = _lib.load()
        self.device = device
        self.stats = _new_stats(device)
        self.scratch = torch.zeros(self.lib.hiprec_scratch_bytes(0), dtype=torch.uint8, device=device)

    def _st(self):
        return _lib.stream_ptr(self.device)

    def reset_clock(self, beta1, beta2):
        _lib.check(self.lib.hiprec_stats_reset(_lib.ptr(self.stats), beta1, beta2, self._st()))

    def gather_rows(self, table, idx):
        out = torch.empty((idx.numel(), table.shape[1]), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.hiprec_gather_rows(
            _lib.ptr(table), table.shape[0], table.shape[1], _lib.ptr(idx), idx.numel(), _lib.ptr(out),
            _lib.ptr(self.stats), self._st()))
        return out

    def scatter_add_rows(self, table, idx, src):
        _lib.check(self.lib.hiprec_scatter_add_rows(
            _lib.ptr(table), table.shape[0], table.shape[1], _lib.ptr(idx), _lib.ptr(src), src.stride(0),
            idx.numel(), _lib.ptr(self.stats), self._st()))

    def grad_on_fetched(self, model, g_flat, fetched, g_fetched, ratings, inv_batch, loss_out):
        """hiprec_ncf_grad with the fetched row buffers standing in for the tables (sample k reads row k): tower /
        head gradients accumulate into ``g_flat`` (laid out like the model's flat buffer), row gradients into the
        zeroed ``g_fetched`` buffers, this rank's share of the loss (scaled by the GLOBAL 1/B) into ``loss_out``."""
        b = ratings.numel()
        base = model.plan(b, g_flat)
        plan = _lib.NcfPlan.from_buffer_copy(base)
        for field in ("user_mlp", "item_mlp", "user_mf", "item_mf"):
            if fetched.get(field) is not None:
                setattr(plan, field, fetched[field].data_ptr())
                setattr(plan, "g_" + field, g_fetched[field].data_ptr())
        plan.n_users = plan.n_items = b
        model.draw_keep_masks(plan, b)
        idx = torch.arange(b, dtype=torch.int64, device=self.device)
        _lib.check(self.lib.hiprec_ncf_grad(
            ctypes.byref(plan), _lib.ptr(idx), _lib.ptr(idx), _lib.ptr(ratings), b, inv_batch, _lib.ptr(self.stats),
            _lib.ptr(self.scratch), self.scratch.numel(), self._st()))
        # d loss / d affine_output.bias travels in the scratch partials: into its gradient slot, the loss into loss_out
        bias_ptr = g_flat.data_ptr() + 4 * model.offset_of("affine_output.bias")
        _lib.check(self.lib.hiprec_finalize_stats(
            _lib.ptr(self.stats), _lib.ptr(self.scratch), ctypes.c_void_p(bias_ptr), _lib.ptr(loss_out), self._st()))

    def advance_clock(self):
        _lib.check(self.lib.hiprec_stats_advance_step(_lib.ptr(self.stats), self._st()))

    def opt_step(self, opt, flat_w, flat_g):
        _lib.check(self.lib.hiprec_opt_dense_step(
            opt.kind, _lib.ptr(flat_w), _lib.ptr(flat_g), _lib.ptr(opt.exp_avg), _lib.ptr(opt.exp_avg_sq),
            flat_w.numel(), opt.lr, opt.beta1, opt.beta2, opt.eps, _lib.ptr(self.stats), None, -1, self._st()))

    def check_status(self):
        st = read_stats(self.stats)
        if st.status:
            clear_status(self.stats)
            raise_on_status(st.status)


_MODELS = {"neumf": "NeuMF", "gmf": "GMF", "mlp": "MLP"}


class ShardedNcfEngine:
    """NeuMF / GMF / MLP with row-sharded embedding tables and a replicated tower; the surface follows the
    single-GPU engines (``train_single_batch(users, items, ratings)`` returns the global-batch loss)."""

    kind = "neumf"

    def __init__(self, config, process_group=None, kernels=None, full_state=None):
        self.config = config
        self.pg = process_group
        self.world = dist.get_world_size(self.pg)
        self.rank = dist.get_rank(self.pg)
        mc = config["model"]
        self.n_users, self.n_items = int(mc["n_users"]), int(mc["n_items"])
        self.device = torch.device(mc["device_str"])
        self.optimizer = HipOptimizer(mc["optimizer"], mc["lr"])
        R, r = self.world, self.rank
        model_cls = getattr(ncf_mod, _MODELS[self.kind])
        if full_state is None:
            # same initial model on every world size: draw the full model like the single-process engine does
            # (same torch seed -> same weights), keep this rank's rows
            with contextlib.redirect_stdout(io.StringIO()):
                full_state = model_cls(dict(mc, device_str="cpu")).state_dict()
        local_cfg = dict(mc, n_users=shard_rows(self.n_users, r, R), n_items=shard_rows(self.n_items, r, R))
        with torch.random.fork_rng(devices=[]), contextlib.redirect_stdout(io.StringIO()):
            self.model = model_cls(local_cfg)
        um, im, ug, ig = self.model._names()
        # (plan field, state_dict name) of the sharded tables, per side
        self._user_tables = [(f, n) for f, n in (("user_mlp", um), ("user_mf", ug)) if n is not None]
        self._item_tables = [(f, n) for f, n in (("item_mlp", im), ("item_mf", ig)) if n is not None]
        self._table_names = {n for _, n in self._user_tables + self._item_tables}
        self.load_full_state_dict(full_state)
        self.model.to(self.device)
        self.k = kernels if kernels is not None else HipNcfKernels(self.device)
        self.k.reset_clock(self.optimizer.beta1 or 0.9, self.optimizer.beta2 or 0.999)
        P = self.model.flat.numel()
        self._g_ext = torch.zeros(P + 2, dtype=torch.float32, device=self.device)   # [gradient | loss share | -]
        self._g_flat = self._g_ext[:P]
        self.optimizer.allocate_state(self.model.flat)
        # everything behind the last table is replicated (tower + head): its gradient is all-reduced with the loss
        self._dense_off = max(self.model.offset_of(n) + self.model.views()[n].numel() for n in self._table_names)
        assert all(self.model.offset_of(n) < self._dense_off for n in self._table_names)
        self.last_loss = float("nan")

    # ---- state ----------------------------------------------------------------------------------------------
    def load_full_state_dict(self, full_state):
        """Keep rows ``rank::world`` of the tables of a full (reference-format) state_dict, and the whole tower."""
        R, r = self.world, self.rank
        local = {}
        for k, v in full_state.items():
            v = torch.as_tensor(v, dtype=torch.float32)
            local[k] = v[r::R] if k in self._table_names else v
        self.model.load_state_dict(local)

    def gather_full_state_dict(self):
        """All-gather the table shards into a reference-compatible state_dict (torch_engine.py:70-73 format)."""
        R = self.world
        full = {}
        for k, v in self.model.state_dict().items():
            if k not in self._table_names:
                full[k] = v.detach().clone()
                continue
            n_total = self.n_users if "user" in k else self.n_items
            rows_max = shard_rows(n_total, 0, R)
            pad = torch.zeros((rows_max,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
            pad[: v.shape[0]] = v
            parts = [torch.empty_like(pad) for _ in range(R)]
            dist.all_gather(parts, pad, group=self.pg)
            out = torch.empty((n_total,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
            for q in range(R):
                out[q::R] = parts[q][: shard_rows(n_total, q, R)]
            full[k] = out
        return full

    # ---- exchange helpers -------------------------------------------------------------------------------------
    def _a2a(self, send, send_counts, recv_counts):
        out = torch.empty((int(sum(recv_counts)),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(out, send.contiguous(), output_split_sizes=list(recv_counts),
                               input_split_sizes=list(send_counts), group=self.pg)
        return out

    def _fetch(self, ids, tables):
        """Rows of ``tables`` (all sharded by the same ids) for this rank's ``ids``: ([b, sum of widths] in batch
        order, routing state for the way back)."""
        R = self.world
        owner = ids % R
        order = torch.argsort(owner, stable=True)
        counts = torch.bincount(owner, minlength=R)
        recv = torch.empty_like(counts)
        dist.all_to_all_single(recv, counts, group=self.pg)
        sc, rc = counts.tolist(), recv.tolist()
        incoming = self._a2a(ids[order], sc, rc)                       # ids of MY rows that peers ask for
        local = torch.div(incoming, R, rounding_mode="floor")
        views = self.model.views()
        payload = torch.cat([self.k.gather_rows(views[n], local) for _, n in tables], dim=1)
        back = self._a2a(payload, rc, sc)                              # rows come back in request order
        fetched = torch.empty_like(back)
        fetched[order] = back
        return fetched, (order, sc, rc, local)

    def _return_grads(self, g_rows, route, tables):
        order, sc, rc, local = route
        gin = self._a2a(g_rows[order], sc, rc)
        g = self.model.views(self._g_flat)
        col = 0
        for _, n in tables:
            w = g[n].shape[1]
            if gin.shape[0] > 0:
                self.k.scatter_add_rows(g[n], local, gin[:, col:col + w].contiguous())
            col += w

    # ---- one step -----------------------------------------------------------------------------------------------
    def train_single_batch(self, users, items, ratings, sync=True):
        """One optimisation step on the GLOBAL batch formed by every rank's samples; returns its BCE loss
        (identical on every rank)."""
        dev, m = self.device, self.model
        users = torch.as_tensor(users, device=dev).to(torch.int64).reshape(-1).contiguous()
        items = torch.as_tensor(items, device=dev).to(torch.int64).reshape(-1).contiguous()
        ratings = torch.as_tensor(ratings, device=dev).to(torch.float32).reshape(-1).contiguous()
        b = users.numel()
        if not (items.numel() == b and ratings.numel() == b):
            raise ValueError("users, items and ratings must be of equal length")
        if b and (int(users.min()) < 0 or int(users.max()) >= self.n_users or int(items.min()) < 0
                  or int(items.max()) >= self.n_items):
            raise IndexError("index out of range in self")
        b_all = torch.tensor([b], dtype=torch.int64, device=dev)
        dist.all_reduce(b_all, group=self.pg)
        B = int(b_all.item())
        if B == 0:
            raise ValueError("empty global batch")
        fu, route_u = self._fetch(users, self._user_tables)
        fi, route_i = self._fetch(items, self._item_tables)
        views = m.views()
        fetched, g_fetched, col = {}, {}, 0
        g_u = torch.zeros_like(fu)
        g_i = torch.zeros_like(fi)
        for src, gsrc, tables in ((fu, g_u, self._user_tables), (fi, g_i, self._item_tables)):
            col = 0
            for field, n in tables:
                w = views[n].shape[1]
                fetched[field] = src[:, col:col + w].contiguous()
                g_fetched[field] = torch.zeros_like(fetched[field])
                col += w
        loss_slot = self._g_ext[-2:]   # hiprec_finalize_stats writes (loss, reg) pairs; NCF has no reg term
        if b > 0:
            self.k.grad_on_fetched(m, self._g_flat, fetched, g_fetched, ratings, 1.0 / B, loss_slot)
        else:
            loss_slot.zero_()
            self.k.advance_clock()   # a rank without samples still ticks the optimizer clock
        for gsrc, tables in ((g_u, self._user_tables), (g_i, self._item_tables)):
            if tables:
                gsrc.copy_(torch.cat([g_fetched[f] for f, _ in tables], dim=1))
        self._return_grads(g_u, route_u, self._user_tables)
        self._return_grads(g_i, route_i, self._item_tables)
        dist.all_reduce(self._g_ext[self._dense_off:], group=self.pg)   # [tower + head gradient | loss]
        loss = self._g_ext[-2].clone()
        self.k.opt_step(self.optimizer, m.flat, self._g_flat)
        if not sync:
            return None
        self.k.check_status()
        self.last_loss = float(loss)
        return self.last_loss

    def train_an_epoch(self, train_loader, epoch_id):
        """Every rank iterates its own share of the (user, item, rating) stream (models/ncf.py:122-139); all loaders
        must yield the same number of batches (one collective step per batch)."""
        total, loss = 0.0, float("nan")
        for batch in train_loader:
            loss = self.train_single_batch(batch[0], batch[1], batch[2])
            total += loss
        if self.rank == 0:
            print("[Training Epoch {}], Loss {}".format(epoch_id, loss))
        return total


class ShardedNeuMFEngine(ShardedNcfEngine):
    kind = "neumf"


class ShardedGMFEngine(ShardedNcfEngine):
    kind = "gmf"


class ShardedMLPEngine(ShardedNcfEngine):
    kind = "mlp"
