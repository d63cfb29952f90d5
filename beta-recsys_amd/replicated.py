"""Data-parallel BPR-MF with REPLICATED tables: the multi-GPU mode for tables that fit one GPU's
caches (BASELINE configs[1]: 2.5 MB of parameters).

Sharding a 2.5 MB model (sharded.py) costs three dependent all-to-alls per 12-µs step; for small
tables it is cheaper to keep a full replica per GPU, let every rank run the fused gradient kernel
on its own batch shard and sum the dense gradients with ONE all-reduce over RCCL/xGMI:

    grad kernel on the local b triples, 1/B with the GLOBAL batch B = world · b
    all-reduce( [ dense gradient | loss part | reg part ] )        (P + 2 floats, one collective)
    identical dense optimizer sweep on every replica

The sum over ranks of per-rank partial gradients (each scaled by the global 1/B) is exactly the
reference's gradient on the concatenated global batch, so the replicas stay bit-identical to each
other and equal to the single-process result up to fp32 summation order (SURVEY.md §8e semantics).
Large tables (BASELINE configs[3]) use the row-sharded engine instead.
"""
import ctypes

import torch
from . import _dist as dist

from . import _lib
from .mf import MFEngine


def allreduce_sum_(buf, group=None):
    """In-place sum of ``buf`` over the process group (RCCL on GPUs, gloo in the CPU tests)."""
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


class DirectAllReduce:
    """``sum_(buf)``: in-place sum of a flat fp32 buffer over the group.  On GPUs through a communicator of the
    engine's own and a direct ncclAllReduce call from ctypes (:mod:`_rccl`; torch.distributed's host path costs
    10-14 us per call, a fifth of a 59 us NCF step) when every rank could make one -- the ranks agree over the torch
    group -- else (gloo, ``dp_collective: "torch"``) through torch.distributed.  Collective on first use."""

    def __init__(self, group, device, mode="rccl"):
        self.pg, self.comm = group, None
        if device.type == "cuda" and mode != "torch" and dist.get_backend(group) == "nccl":
            from . import _rccl

            comm = _rccl.create_communicator(group, device)
            if _rccl.all_ranks_agree(comm is not None, group, device):
                self.comm = comm
            elif comm is not None:
                comm.destroy()

    def sum_(self, buf):
        if self.comm is not None and buf.dtype == torch.float32 and buf.is_contiguous():
            self.comm.all_reduce_sum_(buf)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
        return buf


class ReplicatedMFEngine(MFEngine):
    """``MFEngine`` whose step sums gradients over a process group before the optimizer sweep."""

    _lazy_capable = False   # replicas are for cache-sized tables and sweep a gradient that is dense after the all-reduce

    def __init__(self, config, process_group=None):
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        super().__init__(config)
        # every replica starts from rank 0's weights (the same seed gives the same init anyway)
        if self.model.flat.device.type == "cuda":
            dist.broadcast(self.model.flat, src=0, group=self.pg)

    def _setup(self):
        fresh = not self._buffers_ready
        lib = super()._setup()
        if fresh or self._g_ext.device != self.model.flat.device:
            P = self.model.flat.numel()
            # gradient accumulator with two trailing slots for the (loss, reg) partial sums so that
            # one collective moves everything
            self._g_ext = torch.zeros(P + 2, dtype=torch.float32, device=self.model.flat.device)
            self._g_flat = self._g_ext[:P]
            self._tail = self._g_ext[P:]
            self._gb_ptr = self._g_ext.data_ptr() + 4 * (P - 1)
            self._tail_ptr = self._g_ext.data_ptr() + 4 * P
            self._rows_sgd = False  # replicas always take the dense sweep
            self._dp_ctx = None
            self._fe = None
        self._lib_cached = lib
        return lib

    def _sgd_modes(self):
        return False, False   # replicas always take the dense sweep

    def _step_context(self):
        """hiprec_dp_step with every per-engine constant: the step is then two short C calls around
        the collective (the replicated engine is host-bound whenever the all-reduce is short)."""
        m, opt = self.model, self.optimizer
        key = (m.flat.data_ptr(), self._g_ext.data_ptr(), opt.lr, float(self.reg), self.loss)
        if self._dp_ctx is None or self._dp_ctx[0] != key:
            c = _lib.DpStep()
            c.w, c.g = m.tables(), m.tables(self._g_flat)
            c.stats, c.scratch, c.scratch_bytes = self._stats.data_ptr(), self._scratch.data_ptr(), self._scratch.numel()
            c.loss_reg_out = self._tail_ptr
            c.w_flat, c.g_flat = m.flat.data_ptr(), self._g_flat.data_ptr()
            c.m_flat = None if opt.exp_avg is None else opt.exp_avg.data_ptr()
            c.v_flat = None if opt.exp_avg_sq is None else opt.exp_avg_sq.data_ptr()
            c.n_flat = m.flat.numel()
            c.lr, c.beta1, c.beta2, c.eps = opt.lr, opt.beta1, opt.beta2, opt.eps
            c.reg_coef = float(self.reg)
            c.loss_kind = 0 if self.loss == "bpr" else 1
            c.opt_kind = opt.kind
            self._dp_ctx = (key, c, ctypes.byref(c))
        return self._dp_ctx[2]

    def _enqueue_step(self, batch_data):
        users, a_items, third = self._prepare_batch(batch_data)
        self._enqueue_core(users, a_items, third)

    def enqueue_presorted(self, users, a_items, third):
        """Lean per-step entry for callers that already hold device-resident, contiguous int64
        batches grouped by item (the staged-epoch layout): no conversions, no checks."""
        self._enqueue_core(users, a_items, third)

    def _enqueue_core(self, users, a_items, third):
        """Two C calls around ONE all-reduce of [gradient | loss | reg]: hiprec_mf_dp_step_begin
        (gradient kernel + reduction of the loss partials into the buffer's tail), the collective,
        hiprec_mf_dp_step_end (dense optimizer sweep).  The epoch sums are kept as per-rank shares in
        hiprec_stats (they are linear) and all-reduced once per epoch."""
        lib = self._setup()
        self._stats_are_global = False
        ctx = self._step_context()
        st = _lib.stream_ptr(self.model.flat.device)
        n = users.numel()
        _lib.check(lib.hiprec_mf_dp_step_begin(
            ctx, users.data_ptr(), a_items.data_ptr(), third.data_ptr(), n, 1.0 / (n * self.world), st))
        allreduce_sum_(self._g_ext, self.pg)
        _lib.check(lib.hiprec_mf_dp_step_end(ctx, st))

    # ---- resident epochs: ONE launch + ONE all-reduce per step ------------------------------------------
    # The single-GPU resident epoch runs one fused kernel per step (csrc/mf.hip mf_bpr_fused_kernel:
    # the update of step k-1 rides inside the gradient kernel of step k).  Data-parallel replicas can
    # do the same: the launch of step k needs the SUMMED gradient and loss partials of step k-1, so
    # the all-reduce goes between two launches and covers [partials of step k | gradient of step k],
    # which live back to back in one buffer (three of them rotate, like the gradient buffers of the
    # single-GPU epoch).  Per step: one kernel launch and one collective instead of three launches
    # and one collective, and the loss sums in hiprec_stats are global on every rank.

    def fused_epoch_begin(self):
        """Allocate / clear the rotating buffers and reset the step counter of a fused epoch."""
        lib = self._setup()
        m, opt = self.model, self.optimizer
        if self.loss != "bpr" or m.emb_dim > 256:
            raise RuntimeError("the fused data-parallel epoch covers BPR with emb_dim <= 256")
        dev = m.flat.device
        P = m.flat.numel()
        sf = self._scratch.numel() // 4                    # floats of one scratch block (16-B header first)
        fe = getattr(self, "_fe", None)
        if fe is None or fe["dev"] != dev:
            bufs = [torch.zeros(sf + P, dtype=torch.float32, device=dev) for _ in range(3)]
            fe = {"dev": dev, "bufs": bufs, "w": [m.flat, torch.empty_like(m.flat)],
                  "m": None if opt.exp_avg is None else [opt.exp_avg, torch.empty_like(m.flat)],
                  "v": None if opt.exp_avg_sq is None else [opt.exp_avg_sq, torch.empty_like(m.flat)],
                  # what the collective sums: everything behind the 4-float scratch header
                  "reduce": [b[4:] for b in bufs], "steps": {}}
            self._fe = fe
        else:
            for b in fe["bufs"]:
                b.zero_()
        fe["w"][0] = m.flat
        if fe["m"] is not None:
            fe["m"][0] = opt.exp_avg
        if fe["v"] is not None:
            fe["v"][0] = opt.exp_avg_sq
        fe["steps"] = {}                                   # lr / buffers may have changed since last epoch
        fe["k"], fe["prev_batch"] = 0, 0
        self._stats_are_global = True                      # the fused launches reduce SUMMED partials
        _lib.check(lib.hiprec_stats_begin_epoch(_lib.ptr(self._stats), _lib.stream_ptr(dev)))
        return fe

    def _fused_step_struct(self, k, flush=False):
        """hiprec_fused_step for rotation state k mod 6 (w/m/v ping-pong x gradient buffers mod 3).
        The flush (sweep-only, one thread per element) may update in place: it always writes into
        the engine's own buffers, so nothing is copied back whatever the parity of the epoch."""
        fe = self._fe
        key = (k % 6, flush)
        cached = fe["steps"].get(key)
        if cached is not None:
            return cached[1]
        m, opt = self.model, self.optimizer
        sf = self._scratch.numel() // 4
        c = _lib.FusedStep()
        c.kind, c.dim, c.n_users, c.n_items = opt.kind, m.emb_dim, m.n_users, m.n_items
        out = 0 if flush else (k + 1) & 1
        c.w_read, c.w_write = fe["w"][k & 1].data_ptr(), fe["w"][out].data_ptr()
        if fe["m"] is not None:
            c.m_read, c.m_write = fe["m"][k & 1].data_ptr(), fe["m"][out].data_ptr()
        if fe["v"] is not None:
            c.v_read, c.v_write = fe["v"][k & 1].data_ptr(), fe["v"][out].data_ptr()
        prev, cur, nxt = fe["bufs"][(k + 2) % 3], fe["bufs"][k % 3], fe["bufs"][(k + 1) % 3]
        c.scratch_prev, c.g_prev = prev.data_ptr(), prev.data_ptr() + 4 * sf
        c.scratch_cur, c.g_cur = cur.data_ptr(), cur.data_ptr() + 4 * sf
        c.g_zero = nxt.data_ptr() + 4 * sf
        c.lr, c.beta1, c.beta2, c.eps = opt.lr, opt.beta1, opt.beta2, opt.eps
        c.reg_coef = float(self.reg)
        fe["steps"][key] = (c, ctypes.byref(c))
        return fe["steps"][key][1]

    def fused_step(self, users, pos, neg):
        """One step of a fused epoch on device-resident, contiguous int64 index tensors."""
        self.fused_step_ptr(users.data_ptr(), pos.data_ptr(), neg.data_ptr(), users.numel())

    def fused_step_ptr(self, users_ptr, pos_ptr, neg_ptr, n):
        """Same, on raw device addresses of n consecutive int64 indices each (a slice of a staged epoch
        without creating tensor views: the step is host-bound)."""
        fe = self._fe
        k = fe["k"]
        _lib.check(self._lib_cached.hiprec_mf_bpr_fused_step(
            self._fused_step_struct(k), users_ptr, pos_ptr, neg_ptr, n, fe["prev_batch"],
            1.0 / (n * self.world), self._stats.data_ptr(), _lib.stream_ptr(fe["dev"])))
        allreduce_sum_(fe["reduce"][k % 3], self.pg)
        fe["k"], fe["prev_batch"] = k + 1, n

    def fused_epoch_end(self):
        """Apply the last pending update (sweep-only launch) and hand the state back to the engine's
        own buffers; leaves every rotating buffer clean."""
        fe = self._fe
        k = fe["k"]
        # the flush clears the gradient it applies and marks both scratch blocks empty
        _lib.check(self._lib_cached.hiprec_mf_bpr_fused_step(
            self._fused_step_struct(k, flush=True), None, None, None, 0, fe["prev_batch"], 0.0,
            self._stats.data_ptr(), _lib.stream_ptr(fe["dev"])))
        fe["k"], fe["prev_batch"] = 0, 0

    def _sync_stats(self):
        """Global (all-reduced) loss / reg of the last step replace the local shares in stats."""
        comm = getattr(self, "_direct_comm", None)
        if comm is not None and hasattr(comm, "wait"):   # the C driver's ncclAllReduce calls: a bounded wait (_rccl.py)
            comm.wait(torch.cuda.current_stream(self.model.flat.device),
                      float(self.config["model"].get("collective_timeout_s", 300.0)), "the epoch's all-reduces")
        st = super()._sync_stats()
        if not getattr(self, "_stats_are_global", False):
            loss, reg = (float(x) for x in self._tail.cpu())
            st.loss, st.reg = loss, reg
        return st

    def prepare_epoch(self, train_loader):
        """Replicas iterate their loader batch by batch (one collective per step)."""
        return None

    def _check_equal_batches(self, n, bs):
        """Every step is one collective and scales by 1/(n * world) with the LOCAL n: all ranks must hold the
        same number of triples and use the same batch size (checked once per loader shape, one host sync)."""
        key = (n, bs)
        if getattr(self, "_checked_shape", None) == key:
            return
        t = torch.tensor([n, -n, bs, -bs], dtype=torch.int64, device=self.model.flat.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
        hi_n, lo_n, hi_b, lo_b = (int(x) for x in t.cpu())
        if hi_n != -lo_n or hi_b != -lo_b:
            raise ValueError(
                f"data-parallel replicas need the same local epoch on every rank: this rank has {n} triples in "
                f"batches of {bs}, the group spans {-lo_n}..{hi_n} triples / batches of {-lo_b}..{hi_b}")
        self._checked_shape = key

    def run_resident_epoch(self, train_loader, steps=None):
        """Enqueue one whole epoch of a device-resident loader without reading anything back: the staging of
        this rank's share (MFEngine's device batcher, taken from the side stream when the previous epoch
        prefetched it), the prefetch of the next epoch's staging, one fused launch + one all-reduce per
        step, and the flush.  ``steps=(a, b)``: only steps [a, b) of the epoch (a = 0 stages it, the piece
        that reaches the last step flushes; pieces must be consecutive).  Returns False when the loader
        cannot run resident (caller iterates it instead)."""
        self._setup()
        if self.loss != "bpr" or self.model.emb_dim > 256 or isinstance(train_loader, (list, tuple)):
            return False
        if steps is None or steps[0] == 0:
            staged = MFEngine.prepare_epoch(self, train_loader)   # this rank's share, laid out in visiting order
            if staged is None or staged[3] is not None:
                return False
            self._check_equal_batches(staged[0].numel(), staged[4])
            self._ev_epoch_begin = torch.cuda.Event()
            self._ev_epoch_begin.record(torch.cuda.current_stream(self.model.flat.device))
            self._last_staged = staged   # keeps the arrays alive until the next epoch replaces them
            if self.config["model"].get("prefetch_epoch", True):
                self.prefetch_epoch(train_loader)   # next epoch's staging, on the side stream, while this one runs
            self.fused_epoch_begin()
        users, pos, neg, _, bs = self._last_staged
        n = users.numel()
        n_steps = (n + bs - 1) // bs
        a, b = (0, n_steps) if steps is None else steps
        if not (0 <= a <= b <= n_steps) or a != self._fe["k"]:
            raise ValueError(f"steps {steps}: pieces of an epoch must be consecutive (next step {self._fe['k']} of {n_steps})")
        pu, pp, pn = users.data_ptr(), pos.data_ptr(), neg.data_ptr()
        comm = self._direct_communicator()
        if comm is not None:
            # the whole piece -- per step the fused launch and the all-reduce of its [partials | gradient] -- is
            # enqueued by ONE C call that invokes ncclAllReduce itself (through torch.distributed the step is
            # host-bound: 20-26 us at world size 1 for an 11.5 us kernel)
            fe, m, opt = self._fe, self.model, self.optimizer
            arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])  # noqa: E731
            final = ctypes.c_int32(0)
            _lib.check(self._lib_cached.hiprec_mf_bpr_dp_epoch_fused_range(
                opt.kind, arr(fe["w"]), arr(fe["bufs"]), self._scratch.numel() // 4,
                None if fe["m"] is None else arr(fe["m"]), None if fe["v"] is None else arr(fe["v"]),
                m.flat.numel(), m.n_users, m.n_items, m.emb_dim, pu, pp, pn, n, bs, a, b, self.world, float(self.reg),
                opt.lr, opt.beta1, opt.beta2, opt.eps, self._stats.data_ptr(), comm.all_reduce_fn, comm.comm,
                ctypes.byref(final), _lib.stream_ptr(fe["dev"])))
            if b == n_steps:
                fe["k"], fe["prev_batch"] = 0, 0
            else:
                fe["k"], fe["prev_batch"] = b, min(bs, n - (b - 1) * bs) if b > 0 else 0
            return True
        for k in range(a, b):
            off = k * bs
            self.fused_step_ptr(pu + 8 * off, pp + 8 * off, pn + 8 * off, min(bs, n - off))
        if b == n_steps:
            self.fused_epoch_end()
        return True

    def _direct_communicator(self):
        """The engine's own RCCL communicator (``_rccl.Communicator``) for the C epoch driver, or None: CPU / gloo
        groups, ``config["model"]["dp_collective"] == "torch"``, or any rank that could not create one (the ranks
        agree over the torch group, so either all take the C driver or none).  Collective on first use."""
        if getattr(self, "_direct_comm_tried", False):
            return self._direct_comm
        self._direct_comm_tried, self._direct_comm = True, None
        dev = self.model.flat.device
        if dev.type != "cuda" or self.config["model"].get("dp_collective", "rccl") == "torch":
            return None
        if dist.get_backend(self.pg) != "nccl":
            return None
        from . import _rccl

        comm = _rccl.create_communicator(self.pg, dev)
        if _rccl.all_ranks_agree(comm is not None, self.pg, dev):
            self._direct_comm = comm
        elif comm is not None:
            comm.destroy()
        return self._direct_comm

    def train_an_epoch(self, train_loader, epoch_id):
        lib = self._setup()
        dev = self.model.flat.device
        if self.run_resident_epoch(train_loader):
            # resident loader: every rank must hold the same number of batches (one collective per step), as
            # with any data-parallel loader
            st = self._sync_stats()
            total_loss, total_reg = st.loss_sum, st.reg_sum  # already global
        else:
            _lib.check(lib.hiprec_stats_begin_epoch(_lib.ptr(self._stats), _lib.stream_ptr(dev)))
            for batch_data in train_loader:
                self._enqueue_step(batch_data)
            st = self._sync_stats()
            sums = torch.tensor([st.loss_sum, st.reg_sum], dtype=torch.float64, device=dev)
            allreduce_sum_(sums, self.pg)  # per-rank shares of every step's loss / reg
            total_loss, total_reg = (float(x) for x in sums.cpu())
        if self.rank == 0:
            print(f"[Training Epoch {epoch_id}], Loss {st.loss}, Regularizer {total_reg}")
        self.writer.add_scalar("model/loss", total_loss, epoch_id)
        self.writer.add_scalar("model/regularizer", total_reg, epoch_id)


# ---- the NCF family, data-parallel ---------------------------------------------------------------------
# "tower weights replicated + gradient all-reduce — classic DP for the dense part".  The
# NCF tables of the shipped configs (ML-1M shape, a few MB) are replicated as well: every rank runs
# hiprec_ncf_grad on its share of the global batch with the GLOBAL 1/B, one all-reduce sums
# [flat gradient | loss], an identical dense sweep runs on every replica.

class _ReplicatedNcfMixin:
    """Mix into a ``beta_recsys_amd.ncf`` engine class: ``class E(_ReplicatedNcfMixin, NeuMFEngine)``."""

    def __init__(self, config, process_group=None):
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        super().__init__(config)
        if self.model.flat.device.type == "cuda":
            dist.broadcast(self.model.flat, src=0, group=self.pg)

    def _setup(self):
        fresh = not self._ready
        lib = super()._setup()
        if fresh:
            P = self.model.flat.numel()
            # gradient accumulator with two trailing slots for the (loss, -) partial sums: one collective
            self._g_ext = torch.zeros(P + 2, dtype=torch.float32, device=self.model.flat.device)
            self._g_flat = self._g_ext[:P]
            self._tail = self._g_ext[P:]
        return lib

    def _enqueue_step(self, users, items, ratings):
        lib = self._setup()
        m, opt = self.model, self.optimizer
        dev = m.flat.device
        users = torch.as_tensor(users, device=dev).to(torch.int64).reshape(-1).contiguous()
        items = torch.as_tensor(items, device=dev).to(torch.int64).reshape(-1).contiguous()
        ratings = torch.as_tensor(ratings, device=dev).to(torch.float32).reshape(-1).contiguous()
        B = users.numel()
        if not (items.numel() == B and ratings.numel() == B) or B == 0:
            raise ValueError("users, items and ratings must be non-empty and of equal length")
        st = _lib.stream_ptr(dev)
        plan = m.plan(B, self._g_flat)
        m.draw_keep_masks(plan, B)     # every rank draws its own share's masks (its own rows of the batch)
        _lib.check(lib.hiprec_ncf_grad(
            ctypes.byref(plan), _lib.ptr(users), _lib.ptr(items), _lib.ptr(ratings), B, 1.0 / (B * self.world),
            _lib.ptr(self._stats), _lib.ptr(self._scratch), self._scratch.numel(), st))
        # d loss / d affine_output.bias travels in the scratch partials: put it into its gradient slot and
        # the loss into the tail before the collective
        bias_ptr = self._g_ext.data_ptr() + 4 * m.offset_of("affine_output.bias")
        _lib.check(lib.hiprec_finalize_stats(
            _lib.ptr(self._stats), _lib.ptr(self._scratch), ctypes.c_void_p(bias_ptr),
            ctypes.c_void_p(self._tail.data_ptr()), st))
        self._collective().sum_(self._g_ext)
        _lib.check(lib.hiprec_opt_dense_step(
            opt.kind, _lib.ptr(m.flat), _lib.ptr(self._g_flat), _lib.ptr(opt.exp_avg),
            _lib.ptr(opt.exp_avg_sq), m.flat.numel(), opt.lr, opt.beta1, opt.beta2, opt.eps,
            _lib.ptr(self._stats), None, -1, st))

    def _collective(self):
        if getattr(self, "_dar", None) is None:
            self._dar = DirectAllReduce(self.pg, self.model.flat.device, self.config["model"].get("dp_collective", "rccl"))
        return self._dar

    def _sync_stats(self):
        """The global (all-reduced) loss of the last step replaces this rank's share."""
        st = super()._sync_stats()
        st.loss = float(self._tail[0].item())
        return st


def replicated_ncf_engine(engine_cls):
    """Data-parallel variant of NeuMFEngine / GMFEngine / MLPEngine: ``replicated_ncf_engine(NeuMFEngine)(config)``."""
    return type("Replicated" + engine_cls.__name__, (_ReplicatedNcfMixin, engine_cls), {})


# ---- every other engine that trains: LightGCN, NGCF, PairwiseGMF, Triple2vec, data-parallel -----------------------
# The reference trains all of them on one device (models/lightgcn.py:119-152, ngcf.py:118-149, pairwise_gmf.py:82-116,
# triple2vec.py:94-104); SURVEY 8e: full-graph propagation per step => replicas, no graph partitioning.  A replica
# works on its share of the GLOBAL batch (the reference's batch: `batch_size` in the config stays the global one, which
# is also what NGCF's regulariser and Triple2vec's loss divide by): batch means are scaled by 1 / (local batch x world)
# (FlatModelEngine._batch_share), ONE all-reduce sums [flat gradient | loss share], an identical sweep runs on every
# replica.  Every replica must drop the same edges / messages in a step: the same dropout seed on every rank (the
# counter-based device generator is a function of (seed, step, element); the CPU-replay generators are seeded alike).
# PairwiseGMF: lambda ||v|| is added on rank 0 only and clip_grad_norm_ runs AFTER the reduction, on the global
# gradient, as the reference clips the gradient of the whole batch.

class _ReplicatedFlatMixin:
    """Mix into a FlatModelEngine subclass: ``class E(_ReplicatedFlatMixin, LightGCNEngine)``."""

    def __init__(self, config, process_group=None):
        self.pg = process_group
        self.world = self._dp_world = dist.get_world_size(process_group)
        self.rank = self._dp_rank = dist.get_rank(process_group)
        super().__init__(config)
        if self.model.flat.device.type == "cuda":
            src = dist.get_global_rank(self.pg, 0) if self.pg is not None else 0
            dist.broadcast(self.model.flat, src=src, group=self.pg)

    def _setup(self):
        fresh = not self._ready
        lib = super()._setup()
        if fresh:
            P = self.model.flat.numel()
            self._g_ext = torch.zeros(P + 2, dtype=torch.float32, device=self.model.flat.device)
            self._g_flat = self._g_ext[:P]       # [gradient | loss share | -]: one collective moves both
            self._tail = self._g_ext[P:]
            self._dar = DirectAllReduce(self.pg, self.model.flat.device, self.config["model"].get("dp_collective", "rccl"))
        return lib

    def _enqueue_step(self, batch_data):
        lib = self._setup()
        clips = hasattr(self, "_clip_ws")          # PairwiseGMF: clip the GLOBAL gradient, after the reduction
        if clips:
            self._enqueue_grad(batch_data, clip=False)
        else:
            self._enqueue_grad(batch_data)
        st = _lib.stream_ptr(self.model.flat.device)
        _lib.check(lib.hiprec_finalize_stats(_lib.ptr(self._stats), _lib.ptr(self._scratch), None,
                                             ctypes.c_void_p(self._tail.data_ptr()), st))
        self._dar.sum_(self._g_ext)
        if clips:
            _lib.check(lib.hiprec_clip_grad_norm(
                _lib.ptr(self._g_flat), self._g_flat.numel(), float(self.config["grad_clip"]),
                _lib.ptr(self._clip_ws), self._clip_ws.numel() * 8, st))
        self._enqueue_opt(fold_partials=False)

    def enqueue_epoch(self, *cols):
        """The C epoch drivers of PairwiseGMF / Triple2vec have no collective inside: a data-parallel epoch is the
        loop over this rank's batches (``cols``: the epoch's index columns, first dimension = triples)."""
        lib = self._setup()
        _lib.check(lib.hiprec_stats_begin_epoch(_lib.ptr(self._stats), _lib.stream_ptr(self.model.flat.device)))
        B, n = int(getattr(self, "batch_size", None) or self.model.batch_size), cols[0].shape[0]
        b_local = max(B // self.world, 1)      # the configured batch size is the GLOBAL one
        for k in range(0, n, b_local):
            self._enqueue_step(tuple(c[k:k + b_local] for c in cols))

    def _sync_stats(self):
        """Loss of the last step and the epoch sum as GLOBAL values (this rank's stats hold its shares)."""
        st = super()._sync_stats()
        st.loss = float(self._tail[0].item())
        sums = torch.tensor([st.loss_sum], dtype=torch.float64, device=self.model.flat.device)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.pg)
        st.loss_sum = float(sums[0].item())
        return st


def replicated_flat_engine(engine_cls):
    """Data-parallel variant of LightGCNEngine / NGCFEngine / PairwiseGMFEngine / Triple2vecEngine:
    ``replicated_flat_engine(LightGCNEngine)(config, process_group)``."""
    return type("Replicated" + engine_cls.__name__, (_ReplicatedFlatMixin, engine_cls), {})
