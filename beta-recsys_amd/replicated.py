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
import torch.distributed as dist

from . import _lib
from .mf import MFEngine


def allreduce_sum_(buf, group=None):
    """In-place sum of ``buf`` over the process group (RCCL on GPUs, gloo in the CPU tests)."""
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


class ReplicatedMFEngine(MFEngine):
    """``MFEngine`` whose step sums gradients over a process group before the optimizer sweep."""

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
        return lib

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
        ctx = self._step_context()
        st = _lib.stream_ptr(self.model.flat.device)
        n = users.numel()
        _lib.check(lib.hiprec_mf_dp_step_begin(
            ctx, users.data_ptr(), a_items.data_ptr(), third.data_ptr(), n, 1.0 / (n * self.world), st))
        allreduce_sum_(self._g_ext, self.pg)
        _lib.check(lib.hiprec_mf_dp_step_end(ctx, st))

    def _sync_stats(self):
        """Global (all-reduced) loss / reg of the last step replace the local shares in stats."""
        st = super()._sync_stats()
        loss, reg = (float(x) for x in self._tail.cpu())
        st.loss, st.reg = loss, reg
        return st

    def prepare_epoch(self, train_loader):
        """Replicas iterate their loader batch by batch (one collective per step)."""
        return None

    def train_an_epoch(self, train_loader, epoch_id):
        lib = self._setup()
        dev = self.model.flat.device
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
