"""Plumbing shared by the engines whose model keeps every parameter in ONE flat fp32 buffer (``_FlatModel``)
and whose step is ``<model>_grad`` + ``hiprec_opt_dense_step``: LightGCN, NGCF, PairwiseGMF, Triple2vec.

Nothing here has a counterpart in the reference (it has no such layer): the subclasses mirror
``beta_rec.models.*Engine``; this base only owns the device-side step state (dense gradient, optimizer
moments, ``hiprec_stats``, scratch) and the three things every one of them does with it.
"""
import torch

from . import _lib
from .mf import _new_stats, clear_status, raise_on_status, read_stats
from .torch_engine import ModelEngine


class FlatModelEngine(ModelEngine):
    """Subclasses implement ``_enqueue_grad(batch)`` (zero_grad + forward + loss + backward into
    ``self._g_flat``, loss partials into ``self._scratch``)."""

    _ready = False
    # data-parallel replicas (replicated.replicated_flat_engine) set these: every rank works on its share of the
    # global batch, and a batch MEAN becomes 1 / (local batch x world) so that the sum over ranks is the reference's
    # gradient on the whole batch
    _dp_world = 1
    _dp_rank = 0

    def _batch_share(self):
        return 1.0 / self._dp_world

    def _alloc_extra(self, lib, device):
        """Hook: model-specific workspaces (called once per device, after the common buffers exist)."""

    def _setup(self):
        lib = self.require_hip()
        flat = self.model.flat
        if self._ready and self._g_flat.device == flat.device:
            return lib
        dev = flat.device
        self._g_flat = torch.zeros_like(flat)
        self.optimizer.allocate_state(flat)
        self._scratch = torch.zeros(lib.hiprec_scratch_bytes(0), dtype=torch.uint8, device=dev)
        self._stats = _new_stats(dev, self.optimizer.beta1 or 0.9, self.optimizer.beta2 or 0.999)
        self._alloc_extra(lib, dev)
        self._ready = True
        return lib

    def _sweep_floats(self):
        """How many leading floats of the flat buffers the optimizer moves (all of them by default)."""
        return self.model.flat.numel()

    def _enqueue_opt(self, fold_partials=True):
        """optimizer.step(): the dense sweep, which also folds the loss partials into the stats (unless the caller
        has done that already) and leaves the gradient cleared."""
        lib, m, opt = _lib.load(), self.model, self.optimizer
        _lib.check(lib.hiprec_opt_dense_step(
            opt.kind, _lib.ptr(m.flat), _lib.ptr(self._g_flat), _lib.ptr(opt.exp_avg),
            _lib.ptr(opt.exp_avg_sq), self._sweep_floats(), opt.lr, opt.beta1, opt.beta2, opt.eps,
            _lib.ptr(self._stats), _lib.ptr(self._scratch) if fold_partials else None, -1,
            _lib.stream_ptr(m.flat.device)))

    def _enqueue_step(self, batch_data):
        self._enqueue_grad(batch_data)
        self._enqueue_opt()

    def _sync_stats(self):
        """The one host sync of a step / epoch; out-of-range ids surface here as IndexError (the sticky
        status word is cleared and a partially accumulated gradient dropped, so the engine stays usable)."""
        st = read_stats(self._stats)
        if st.status:
            clear_status(self._stats)
            self._g_flat.zero_()
            raise_on_status(st.status)
        return st

    def _finish_backward_only(self):
        """After ``_enqueue_grad``: reduce the loss partials without an optimizer call, hand out a copy of
        the gradient and clear it.  Returns ``(stats, grads)``."""
        lib = _lib.load()
        _lib.check(lib.hiprec_finalize_stats(_lib.ptr(self._stats), _lib.ptr(self._scratch), None, None,
                                             _lib.stream_ptr(self.model.flat.device)))
        st = self._sync_stats()
        grads = {k: v.clone() for k, v in self.model.views(self._g_flat).items()}
        self._g_flat.zero_()
        return st, grads

    def backward_only(self, batch_data):
        """zero_grad + forward + loss + backward without the optimizer step: ``(loss, grads)``."""
        self._enqueue_grad(batch_data)
        st, grads = self._finish_backward_only()
        return st.loss, grads

    def load_optimizer_state(self, step, exp_avg=None, exp_avg_sq=None):
        """Resume from a reference optimizer state: step count + per-parameter moment dicts keyed like
        ``state_dict`` (``exp_avg_sq`` doubles as RMSprop's ``square_avg``); ``None`` zeroes a moment."""
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
