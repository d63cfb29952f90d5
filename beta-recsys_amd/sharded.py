"""Row-sharded BPR-MF training across the GPUs of one node.

The reference is single-device only (no ``torch.distributed`` anywhere); this is the MI355X-native
scale-out of the same step.  One process per GPU, ``torch.distributed`` over RCCL/xGMI (backend
"nccl"); the CPU tests drive the same host code over gloo with the oracle standing in for the HIP
kernels.

Partitioning:  ``owner(row) = row mod R`` for BOTH tables (and their bias tables); local index
``row // R``.  ``global_bias`` is replicated and its gradient all-reduced.

One step on a global batch (every rank contributes its local batch ``b_r``; ``1/B`` uses the global
``B`` so the result equals the single-process reference on the concatenated batch up to fp32
summation order):

    A2A-1  triples -> owner(user)                           24 B / triple
    A2A-2  item ids -> owner(item), rows (+bias) back       2·(D+1)·4 B / triple
    step   the single-GPU gradient kernel on (local user shard, fetched item rows)
    A2A-3  item-row gradients -> owner(item), scatter-add   2·(D+1)·4 B / triple
    all-reduce of (loss, reg, d loss / d global_bias)       3 floats
    local dense optimizer sweep over the shard

xGMI is point-to-point (7 links per GPU): an all-to-all drives all links at once, which is why the
exchange is expressed as all-to-all rather than ring collectives.
"""
import contextlib
import ctypes
import os
import io

import torch
from . import _dist as dist

from . import _lib
from .mf import MF, _new_stats, clear_status, raise_on_status, read_stats
from .torch_engine import HipOptimizer

KEYS = ("global_bias", "user_emb.weight", "item_emb.weight", "user_bias.weight", "item_bias.weight")


def shard_rows(n_rows, rank, world):
    """Number of rows owned by ``rank`` under owner(row) = row mod world."""
    return (n_rows - rank + world - 1) // world


class HipKernels:
    """The product compute backend: libhiprec through the C ABI (no fallback)."""

    def __init__(self, device):
        if device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError(
                "ShardedMFEngine computes through libhiprec.so on an MI355X; there is deliberately "
                f"no CPU fallback (device {device})")
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
            _lib.ptr(table), table.shape[0], table.shape[1], _lib.ptr(idx), idx.numel(),
            _lib.ptr(out), _lib.ptr(self.stats), self._st()))
        return out

    def route_bucket(self, keys, n_dest, cap):
        """slot[k] = dest*cap + arrival position (dest = key mod n_dest), -1 for padding/overflow."""
        slots = torch.empty_like(keys)
        counts = torch.empty(n_dest, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.hiprec_route_bucket(
            _lib.ptr(keys), keys.numel(), n_dest, cap, _lib.ptr(counts), _lib.ptr(slots),
            _lib.ptr(self.stats), self._st()))
        return slots

    def scatter_add_rows(self, table, idx, src):
        _lib.check(self.lib.hiprec_scatter_add_rows(
            _lib.ptr(table), table.shape[0], table.shape[1], _lib.ptr(idx), _lib.ptr(src),
            src.stride(0), idx.numel(), _lib.ptr(self.stats), self._st()))

    def bpr_grad(self, w, g, users, pos, neg, inv_batch, reg_coef):
        """w / g: dicts of tensors keyed like state_dict (item tables = the fetched rows).
        Returns a float32 tensor [loss_part, reg_part, d_global_bias_part] on the device."""
        def tables(t):
            return _lib.MfTables(
                t["user_emb.weight"].data_ptr(), t["item_emb.weight"].data_ptr(),
                t["user_bias.weight"].data_ptr(), t["item_bias.weight"].data_ptr(),
                t["global_bias"].data_ptr(), t["user_emb.weight"].shape[0],
                t["item_emb.weight"].shape[0], t["user_emb.weight"].shape[1], 0)

        wt, gt = tables(w), tables(g)
        gb_part = torch.zeros(1, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.hiprec_mf_bpr_grad(
            ctypes.byref(wt), ctypes.byref(gt), _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), None,
            users.numel(), inv_batch, reg_coef, _lib.ptr(self.stats), _lib.ptr(self.scratch),
            self.scratch.numel(), self._st()))
        _lib.check(self.lib.hiprec_finalize_stats(
            _lib.ptr(self.stats), _lib.ptr(self.scratch), _lib.ptr(gb_part), None, self._st()))
        head = self.stats[:8].view(torch.float32)  # hiprec_stats.loss, .reg
        return torch.cat([head, gb_part])

    # ---- fused routing / packing of the padded step (csrc/shard.hip): one launch each -----------------
    def _counts(self):
        if getattr(self, "_cnt", None) is None:
            self._cnt = torch.zeros(64, dtype=torch.int32, device=self.device)
        return self._cnt

    def route_triples(self, users, pos, neg, n_dest, cap, send):
        """send[n_dest*cap, 3] = (user, pos, neg) bucketed by owner(user), -1 padding."""
        _lib.check(self.lib.hiprec_shard_route_triples(
            _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), users.numel(), n_dest, cap, _lib.ptr(self._counts()),
            _lib.ptr(send), _lib.ptr(self.stats), self._st()))

    def route_items(self, recv, n_dest, cap, req, slot_pos, slot_neg, u_loc):
        """From received triples: item request buffer + the slots their rows come back in + local users."""
        _lib.check(self.lib.hiprec_shard_route_items(
            _lib.ptr(recv), recv.shape[0], n_dest, cap, _lib.ptr(self._counts()), _lib.ptr(req),
            _lib.ptr(slot_pos), _lib.ptr(slot_neg), _lib.ptr(u_loc), _lib.ptr(self.stats), self._st()))

    def gather_payload(self, item_emb, item_bias, incoming, n_dest, payload, local_idx):
        """payload[k] = [item_emb row | item_bias] of incoming[k] // n_dest (zeros for padding)."""
        _lib.check(self.lib.hiprec_shard_gather_payload(
            _lib.ptr(item_emb), _lib.ptr(item_bias), item_emb.shape[0], item_emb.shape[1], _lib.ptr(incoming),
            incoming.numel(), n_dest, _lib.ptr(payload), _lib.ptr(local_idx), _lib.ptr(self.stats), self._st()))

    def split_rows(self, src, emb, bias):
        _lib.check(self.lib.hiprec_shard_split_rows(_lib.ptr(src), src.shape[0], emb.shape[1], _lib.ptr(emb),
                                                    _lib.ptr(bias), self._st()))

    def join_rows(self, emb, bias, dst):
        _lib.check(self.lib.hiprec_shard_join_rows(_lib.ptr(emb), _lib.ptr(bias), emb.shape[0], emb.shape[1],
                                                   _lib.ptr(dst), self._st()))

    def bpr_grad_into(self, w, g, users, pos, neg, inv_batch, reg_coef, part):
        """bpr_grad writing [loss, reg, d global_bias] into the caller's 3-float tensor (no allocation)."""
        def tables(t):
            return _lib.MfTables(
                t["user_emb.weight"].data_ptr(), t["item_emb.weight"].data_ptr(),
                t["user_bias.weight"].data_ptr(), t["item_bias.weight"].data_ptr(),
                t["global_bias"].data_ptr(), t["user_emb.weight"].shape[0],
                t["item_emb.weight"].shape[0], t["user_emb.weight"].shape[1], 0)

        wt, gt = tables(w), tables(g)
        part.zero_()
        _lib.check(self.lib.hiprec_mf_bpr_grad(
            ctypes.byref(wt), ctypes.byref(gt), _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), None,
            users.numel(), inv_batch, reg_coef, _lib.ptr(self.stats), _lib.ptr(self.scratch),
            self.scratch.numel(), self._st()))
        _lib.check(self.lib.hiprec_finalize_stats(
            _lib.ptr(self.stats), _lib.ptr(self.scratch), ctypes.c_void_p(part.data_ptr() + 8),
            _lib.ptr(part), self._st()))

    def sgd_rows(self, model, g_flat, users, items, lr, user_stamp, item_stamp, stamp):
        """Exact SGD on the rows named by `users` / `items` only (-1 = padding): w -= lr * g, g = 0, plus the
        scalar bias.  Plain SGD leaves untouched rows bit-identical, so a shard need not sweep its slice."""
        w, g = model.tables(), model.tables(g_flat)
        n = users.numel() if users is not None else items.numel()
        _lib.check(self.lib.hiprec_mf_sgd_rows(
            ctypes.byref(w), ctypes.byref(g), _lib.ptr(users), _lib.ptr(items), None, None, n, lr,
            _lib.ptr(user_stamp), _lib.ptr(item_stamp), stamp, _lib.ptr(self.stats), None, self._st()))

    # ---- the epoch planner (csrc/plan.hip) and the planned step (csrc/mf_owned.hip REMOTE variants, csrc/shard.hip) ----
    def _i32(self, *shape):
        return torch.empty(shape, dtype=torch.int32, device=self.device)

    def plan_route(self, users, pos, neg, perm, bs, world, n_users, n_items):
        """-> (send int32 [n, 3]: (user // world, pos, neg), (destination, step)-ordered; cnt_ds int32 [world, S])."""
        n = users.numel()
        S = max((n + bs - 1) // bs, 1)
        tiles = self.lib.hiprec_plan_route_tiles(n, bs)
        ws, cnt_ds, send = self._i32(max(world * tiles, 1)), self._i32(world, S), self._i32(max(n, 1), 3)
        _lib.check(self.lib.hiprec_plan_route_triples(
            _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), _lib.ptr(perm) if perm is not None else None, n, bs, world,
            n_users, n_items, _lib.ptr(ws), _lib.ptr(cnt_ds), _lib.ptr(send), _lib.ptr(self.stats), self._st()))
        return send[:n], cnt_ds

    def plan_status(self):
        """The status word of the device stats as a 1-element int64 tensor (rides in the plan's host read)."""
        off = _lib.Stats.status.offset
        return self.stats[off:off + 4].view(torch.int32).to(torch.int64)

    def plan_place_triples(self, recv, recv_cnt, S, cap):
        R = recv_cnt.shape[0]
        U, P, N = (torch.empty(S * cap, dtype=torch.int64, device=self.device) for _ in range(3))
        ws = self._i32(2 * R * S + 1 + S)
        _lib.check(self.lib.hiprec_plan_place_triples(
            _lib.ptr(recv), recv.shape[0], _lib.ptr(recv_cnt), R, S, cap, _lib.ptr(ws), _lib.ptr(U), _lib.ptr(P),
            _lib.ptr(N), self._st()))
        return U, P, N, ws[2 * R * S + 1:]      # + triples per step: plan_item_slots pads only behind them

    def plan_item_slots(self, U, P, N, S, cap, world, n_users_local, n_items, fill=None):
        """Ownership tables of the step blocks + slots of the exchange buffers + the blocks grouped by positive item."""
        lib, tot = self.lib, S * cap
        bits = lib.hiprec_ownership_table_bits(cap)
        T = 1 << bits
        keys, own, occ = self._i32(max(lib.hiprec_ownership_ws_ints(tot, cap, bits), 1)), self._i32(3, tot), self._i32(3 * tot)
        total, tab_keys, pos_cnt, slot_of = (self._i32(S, T) for _ in range(4))
        _lib.check(lib.hiprec_batch_row_ownership_tables(
            _lib.ptr(U), _lib.ptr(P), _lib.ptr(N), tot, cap, max(n_users_local, 1), n_items, bits, _lib.ptr(keys),
            _lib.ptr(total), _lib.ptr(own), _lib.ptr(tab_keys), _lib.ptr(pos_cnt), _lib.ptr(occ), self._st()))
        ws = self._i32(lib.hiprec_plan_slot_ws_ints(S, bits, world))
        req_cnt, req_ds, ex_req = self._i32(S, world), self._i32(world, S), self._i32(S, world)
        n_slots, send_base = self._i32(S), self._i32(world * S + 1)
        # at most 2 distinct items per live triple
        req_send = self._i32(2 * tot)
        U2, SP, SN = (torch.empty(tot, dtype=torch.int64, device=self.device) for _ in range(3))
        own2 = self._i32(3, tot)
        slot_stride = 2 * cap + world    # one flag per slot: "several triples of the step add into it"
        slot_shared = torch.empty((S, slot_stride), dtype=torch.uint8, device=self.device)
        _lib.check(lib.hiprec_plan_item_slots(
            _lib.ptr(U), S, cap, world, max(n_users_local, 1), bits, _lib.ptr(own), _lib.ptr(occ), _lib.ptr(tab_keys),
            _lib.ptr(pos_cnt), _lib.ptr(ws), _lib.ptr(slot_of), _lib.ptr(req_cnt), _lib.ptr(req_ds), _lib.ptr(ex_req),
            _lib.ptr(n_slots), _lib.ptr(send_base), _lib.ptr(req_send), _lib.ptr(U2), _lib.ptr(SP), _lib.ptr(SN),
            _lib.ptr(own2), _lib.ptr(total), _lib.ptr(slot_shared), slot_stride,
            _lib.ptr(fill) if fill is not None else None, self._st()))
        return {"U": U2, "SP": SP, "SN": SN, "own": own2, "total": total, "stride": T, "req_cnt": req_cnt,
                "req_ds": req_ds, "ex_req": ex_req, "req_send": req_send, "slot_shared": slot_shared,
                "slot_bound": slot_stride}

    def plan_place_requests(self, incoming, in_qs, S, n_rows_local=0):
        """-> (in_idx, ex_in, dup_bits): dup_bits uint32 [S, words] = rows more than one peer asks for in a step."""
        R, n_in = in_qs.shape[0], incoming.numel()
        ws, in_idx = self._i32(2 * R * S + 1), self._i32(n_in + R * S)
        step_off, ex_in = self._i32(S + 1), self._i32(S, R)
        words = (n_rows_local + 31) // 32
        dup_ws = self._i32(S, max(words, 1)) if words > 0 else None
        _lib.check(self.lib.hiprec_plan_place_requests(
            _lib.ptr(incoming), n_in, _lib.ptr(in_qs), R, S, _lib.ptr(ws), _lib.ptr(in_idx), _lib.ptr(step_off),
            _lib.ptr(ex_in), n_rows_local, _lib.ptr(dup_ws) if dup_ws is not None else None, self._st()))
        return in_idx, ex_in, dup_ws

    def payload_zero(self, item_emb, item_bias, idx, payload, g_send):
        """payload[k] = [item_emb row | item_bias] of LOCAL row idx[k] (zeros for -1); g_send = 0."""
        _lib.check(self.lib.hiprec_shard_payload_zero(
            _lib.ptr(item_emb), _lib.ptr(item_bias), item_emb.shape[0], item_emb.shape[1], _lib.ptr(idx), idx.numel(),
            0, 0, _lib.ptr(payload), None, _lib.ptr(g_send), g_send.numel(), None, _lib.ptr(self.stats), self._st()))

    def owned_remote_step(self, model, fetched, g_send, n_slots, users, slot_pos, slot_neg, own, total, arrived, acc,
                          inv_batch, reg_coef, lr):
        _lib.check(self.lib.hiprec_mf_bpr_owned_remote_step(
            _lib.ptr(model.flat), model.n_users, model.n_items, model.emb_dim, _lib.ptr(fetched), _lib.ptr(g_send),
            n_slots, _lib.ptr(users), _lib.ptr(slot_pos), _lib.ptr(slot_neg), _lib.ptr(own[0]), _lib.ptr(own[1]),
            _lib.ptr(own[2]), _lib.ptr(total), _lib.ptr(arrived), _lib.ptr(acc), users.numel(), inv_batch, reg_coef,
            lr, _lib.ptr(self.stats), _lib.ptr(self.scratch), self._st()))

    def grad_remote_step(self, model, g_flat, fetched, g_send, n_slots, users, slot_pos, slot_neg, own, total,
                         inv_batch, reg_coef):
        """The dense optimizers' form: user-row gradients into g_flat, item-slot gradients into g_send."""
        _lib.check(self.lib.hiprec_mf_bpr_grad_remote_step(
            _lib.ptr(model.flat), _lib.ptr(g_flat), model.n_users, model.n_items, model.emb_dim, _lib.ptr(fetched),
            _lib.ptr(g_send), n_slots, _lib.ptr(users), _lib.ptr(slot_pos), _lib.ptr(slot_neg), _lib.ptr(own[0]),
            _lib.ptr(own[1]), _lib.ptr(own[2]), _lib.ptr(total), users.numel(), inv_batch, reg_coef,
            _lib.ptr(self.stats), _lib.ptr(self.scratch), self._st()))

    def publish_partials(self, g_send, dim, extra_rows):
        _lib.check(self.lib.hiprec_shard_publish_partials(_lib.ptr(self.scratch), _lib.ptr(g_send), dim,
                                                          _lib.ptr(extra_rows.to(torch.int64)), extra_rows.numel(),
                                                          self._st()))

    def apply_finish(self, t_emb, t_bias, idx, g_recv, coef, extra_pos, scalar_target, scalar_coef, first_of_epoch):
        """target row idx[k] += coef * g_recv[k]; the peers' extra rows -> stats, scalar += scalar_coef * its gradient."""
        _lib.check(self.lib.hiprec_shard_apply_finish(
            _lib.ptr(t_emb), _lib.ptr(t_bias), t_emb.shape[0], t_emb.shape[1], _lib.ptr(idx), _lib.ptr(g_recv),
            idx.numel(), 0, 0, None, coef, _lib.ptr(extra_pos), extra_pos.numel(), _lib.ptr(scalar_target), scalar_coef,
            1 if first_of_epoch else 0, None, _lib.ptr(self.stats), self._st()))

    def planned_steps(self, plan, bufs, model, g_flat, opt, a, b, reg, comm, lazy=None, self_exchange=False):
        """Steps [a, b) of a planned epoch -- kernels AND exchanges -- enqueued by ONE C call
        (hiprec_shard_planned_steps_ex); comm: an _rccl.Communicator (None at world size 1).  ``self_exchange``: the
        rank's own segment of both exchanges goes through the communicator too (HIPREC_SHARD_EXCHANGE_SELF)."""
        c = plan.get("_c")
        if c is None:   # the structs and the host arrays they point at live as long as the plan
            import numpy as np

            host = {k: np.ascontiguousarray(plan[k], dtype=np.int64) for k in ("in_off_h", "n_slots_h", "req_cnt_h",
                                                                                "in_cnt_h")}
            sp = _lib.ShardPlan(
                plan["world"], plan["rank"], plan["S"], plan["cap"], plan["bs"], plan["n"], plan["U"].data_ptr(),
                plan["SP"].data_ptr(), plan["SN"].data_ptr(), plan["own"].data_ptr(), plan["total"].data_ptr(),
                plan["stride"], plan["in_idx"].data_ptr(), plan["ex_req"].data_ptr(), plan["ex_in"].data_ptr(),
                host["in_off_h"].ctypes.data, host["n_slots_h"].ctypes.data, host["req_cnt_h"].ctypes.data,
                host["in_cnt_h"].ctypes.data,
                plan["slot_shared"].data_ptr() if plan.get("slot_shared") is not None else None,
                plan["slot_shared"].shape[1] if plan.get("slot_shared") is not None else 0,
                plan["dup_bits"].data_ptr() if plan.get("dup_bits") is not None else None,
                plan["dup_bits"].shape[1] if plan.get("dup_bits") is not None else 0,
                *((pc[0].data_ptr(), pc[1].data_ptr(), pc[3], pc[2].data_ptr()) if (pc := plan.get("contrib")) is not None
                  else (None, None, 0, None)))
            c = plan["_c"] = (sp, host)
        dense = opt.name != "sgd"
        sb = _lib.ShardBufs(
            model.flat.data_ptr(), model.n_users, model.n_items, model.emb_dim, 0, bufs["payload"].data_ptr(),
            bufs["g_recv"].data_ptr(), bufs["fetched"].data_ptr(), bufs["g_send"].data_ptr(),
            bufs["arrived"].data_ptr(), bufs["acc"].data_ptr(), self.scratch.data_ptr(),
            g_flat.data_ptr() if dense else None,
            opt.exp_avg.data_ptr() if opt.exp_avg is not None else None,
            opt.exp_avg_sq.data_ptr() if opt.exp_avg_sq is not None else None,
            lazy["stamp_u"].data_ptr() if lazy else None, lazy["stamp_i"].data_ptr() if lazy else None,
            lazy["scalars"].data_ptr() if lazy else None, lazy["scalars"].shape[0] if lazy else 0,
            bufs["cbuf"].data_ptr() if bufs.get("cbuf") is not None else None,
            bufs["cbias"].data_ptr() if bufs.get("cbias") is not None else None)
        fns = None
        if comm is not None:
            fns = ctypes.byref(_lib.NcclFns(comm.send_fn, comm.recv_fn, comm.group_start_fn, comm.group_end_fn))
        if self_exchange and comm is None:
            raise RuntimeError("shard_self_exchange needs the RCCL binding (a communicator with ncclSend / ncclRecv)")
        _lib.check(self.lib.hiprec_shard_planned_steps_ex(
            ctypes.byref(c[0]), ctypes.byref(sb), a, b, opt.kind, reg, opt.lr, opt.beta1, opt.beta2, opt.eps, fns,
            ctypes.c_void_p(comm.comm.value if comm is not None else None),
            _lib.SHARD_EXCHANGE_SELF if self_exchange else 0, _lib.ptr(self.stats), self._st()))

    # ---- exact lazy Adam / RMSprop (csrc/lazy_opt.hip) -------------------------------------------------------------
    LAZY_SCALARS = _lib.LAZY_SCALARS_CAP     # steps whose bias corrections are tabulated (default betas converge by t ~ 36 800)

    def lazy_state(self, model, g_flat, opt):
        """The per-row stamps (-1 = never touched) and Adam's per-step scalars table of a shard, with the C struct
        that names them next to w / g / m / v."""
        dev = self.device
        lz = {"stamp_u": torch.full((max(model.n_users, 1),), -1, dtype=torch.int32, device=dev),
              "stamp_i": torch.full((max(model.n_items, 1),), -1, dtype=torch.int32, device=dev),
              "scalars": _lib.lazy_scalars_table(opt, dev, self.LAZY_SCALARS), "dirty": False}
        lz["c"] = _lib.LazyState(
            model.flat.data_ptr(), g_flat.data_ptr(), opt.exp_avg.data_ptr() if opt.exp_avg is not None else None,
            opt.exp_avg_sq.data_ptr(), model.n_users, model.n_items, model.emb_dim, opt.kind, lz["stamp_u"].data_ptr(),
            lz["stamp_i"].data_ptr(), lz["scalars"].data_ptr(), self.LAZY_SCALARS, 0, opt.lr, opt.beta1, opt.beta2,
            opt.eps)
        return lz

    @staticmethod
    def _lazy_rows(users, items32):
        return _lib.LazyRows(users.data_ptr(), users.numel(), None, 0, None, 0, items32.data_ptr(), items32.numel())

    def lazy_catchup(self, lz, users, items32):
        rows = self._lazy_rows(users, items32)
        _lib.check(self.lib.hiprec_lazy_catchup(ctypes.byref(lz["c"]), ctypes.byref(rows), _lib.ptr(self.stats), self._st()))

    def lazy_update(self, lz, users, items32):
        rows = self._lazy_rows(users, items32)
        _lib.check(self.lib.hiprec_lazy_update(ctypes.byref(lz["c"]), ctypes.byref(rows), None, _lib.ptr(self.stats),
                                               self._st()))

    def lazy_flush(self, lz):
        _lib.check(self.lib.hiprec_lazy_flush(ctypes.byref(lz["c"]), _lib.ptr(self.stats), self._st()))

    def lazy_mark_current(self, lz):
        _lib.check(self.lib.hiprec_lazy_mark_current(ctypes.byref(lz["c"]), _lib.ptr(self.stats), self._st()))

    def epoch_stats(self):
        """(last loss, last reg, loss sum, reg sum) of the epoch (synchronises)."""
        self.check_status()
        st = read_stats(self.stats)
        return st.loss, st.reg, st.loss_sum, st.reg_sum

    def advance_clock(self):
        """A rank that received no triple this step still has to tick the optimizer clock."""
        _lib.check(self.lib.hiprec_stats_advance_step(_lib.ptr(self.stats), self._st()))

    def opt_step(self, opt, flat_w, flat_g, step):
        # the clock is advanced by bpr_grad (one grad call per step), `step` is informational
        _lib.check(self.lib.hiprec_opt_dense_step(
            opt.kind, _lib.ptr(flat_w), _lib.ptr(flat_g), _lib.ptr(opt.exp_avg),
            _lib.ptr(opt.exp_avg_sq), flat_w.numel(), opt.lr, opt.beta1, opt.beta2, opt.eps,
            _lib.ptr(self.stats), None, -1, self._st()))

    comm_watch = None       # (communicator, timeout in seconds): set by the engine whose C driver posts exchanges

    def wait_exchanges(self):
        """Before the host blocks on anything the steps produced: a wait that cannot hang for ever (``_rccl.Communicator
        .wait``: polls the stream and the communicator's asynchronous error, aborts it after the timeout)."""
        if self.comm_watch is not None:
            comm, timeout = self.comm_watch
            if comm is not None and hasattr(comm, "wait"):
                comm.wait(torch.cuda.current_stream(self.device), timeout, "the planned steps' exchanges")

    def check_status(self):
        self.wait_exchanges()
        st = read_stats(self.stats)
        if st.status:
            clear_status(self.stats)
            raise_on_status(st.status)

    def clear_status(self):
        clear_status(self.stats)


class ShardedMFEngine:
    """BPR-MF with row-sharded tables; the surface follows ``MFEngine`` where it makes sense."""

    def __init__(self, config, process_group=None, kernels=None, full_state=None):
        self.config = config
        self.pg = process_group
        self.world = dist.get_world_size(self.pg)
        self.rank = dist.get_rank(self.pg)
        mc = config["model"]
        self.n_users, self.n_items, self.emb_dim = int(mc["n_users"]), int(mc["n_items"]), int(mc["emb_dim"])
        self.device = torch.device(mc["device_str"])
        self.loss = mc["loss"] if "loss" in mc else "bpr"
        if self.loss != "bpr":
            raise RuntimeError(f"Unsupported loss type {self.loss} for the sharded engine: 'bpr' only")
        self.reg = config["model"]["reg"] if "reg" in config else 0.0  # quirk Q1, as MFEngine
        self.optimizer = HipOptimizer(mc["optimizer"], mc["lr"])
        R, r = self.world, self.rank
        local_cfg = dict(mc)
        local_cfg["n_users"] = shard_rows(self.n_users, r, R)
        local_cfg["n_items"] = shard_rows(self.n_items, r, R)
        local_init = full_state is None and mc.get("shard_init", "full") == "local"
        if full_state is None and not local_init:
            # Same initial model on every world size: draw the full tables like the single-process
            # model does (same torch seed -> same weights), keep this rank's rows.
            with contextlib.redirect_stdout(io.StringIO()):
                full = MF(dict(mc, device_str="cpu"))
            full_state = full.state_dict()
        with torch.random.fork_rng(devices=[]):
            if local_init:
                # `shard_init: "local"`: every rank draws only its own rows (tables too large to draw whole on
                # every rank: configs[3] is 5.7 GB); same distribution, not the single-process stream
                torch.manual_seed(torch.initial_seed() + 7919 * (r + 1))
            with contextlib.redirect_stdout(io.StringIO()):
                self.model = MF(local_cfg)
        if not local_init:
            self.load_full_state_dict(full_state)
        self.model.to(self.device)
        self.k = kernels if kernels is not None else HipKernels(self.device)
        self.k.reset_clock(self.optimizer.beta1 or 0.9, self.optimizer.beta2 or 0.999)
        self._g_flat = torch.zeros_like(self.model.flat)
        self.optimizer.allocate_state(self.model.flat)
        self.step_count = 0
        self.last = (float("nan"), float("nan"))
        # Adam / RMSprop on the planned epoch path: `dense_opt` = "sweep" (every element every step: 28 bytes per
        # parameter per step), "lazy" (exact lazy replay, csrc/lazy_opt.hip: only the step's rows move; flushed before
        # anybody reads the tables) or "auto" (lazy from 64 MB of shard on -- below that the sweep is cache traffic)
        mode = mc["dense_opt"] if "dense_opt" in mc else "auto"
        if mode not in ("sweep", "lazy", "auto"):
            raise ValueError(f"dense_opt must be 'sweep', 'lazy' or 'auto', not {mode!r}")
        self._lazy = None
        if (self.optimizer.name != "sgd" and isinstance(self.k, HipKernels) and self.emb_dim <= 256
                and (mode == "lazy" or (mode == "auto" and self.model.flat.numel() * 4 >= (64 << 20)
                                        and _lib.lazy_betas_converge(self.optimizer)))):
            self._lazy = self.k.lazy_state(self.model, self._g_flat, self.optimizer)
        # "padded": fixed-capacity all-to-alls, bucketing on the device, no host sync per step (all
        # ranks must feed the same local batch size); "variable": exact-size all-to-alls with
        # host-side split sizes (any batch sizes, one host sync per exchange)
        self.routing = mc["routing"] if "routing" in mc else "padded"
        self.route_slack = float(mc["route_slack"]) if "route_slack" in mc else 1.25
        # plain SGD on big shards visits only the rows of the step (as MFEngine does above 64 MB):
        # `sgd_mode` = "rows" | "dense" | "auto"
        mode = mc["sgd_mode"] if "sgd_mode" in mc else "auto"
        self._rows_sgd = self.optimizer.name == "sgd" and self.routing == "padded" and (
            mode == "rows" or (mode == "auto" and self.model.flat.numel() * 4 >= (64 << 20)))
        if self._rows_sgd:
            self._user_stamp = torch.zeros(max(self.model.n_users, 1), dtype=torch.int32, device=self.device)
            self._item_stamp = torch.zeros(max(self.model.n_items, 1), dtype=torch.int32, device=self.device)
            self._stamp = 0

    # ---- state ------------------------------------------------------------------------------
    def load_full_state_dict(self, full_state):
        """Keep rows ``rank::world`` of a full (reference-format) state_dict."""
        R, r = self.world, self.rank
        if getattr(self, "_lazy", None) is not None:
            self.flush_lazy()      # the moments stay: every row must be current before its weights are replaced
        local = {}
        for k in KEYS:
            v = torch.as_tensor(full_state[k], dtype=torch.float32)
            local[k] = v if k == "global_bias" else v[r::R]
        self.model.load_state_dict(local)

    # Lazy Adam / RMSprop: the piece that ends a planned epoch flushes, so whoever reads the shard afterwards
    # (state_dict, predict, the moments) sees what the dense steps would have left.  Not a config key: a run that leaves
    # rows lagging between epochs (tools/exp_planned.py, LAZY_FLUSH=demand) flips this attribute and flushes by hand
    # before it reads anything (the config key made stale shards reachable from product configs).
    flush_lazy_every_epoch = True

    def flush_lazy(self):
        """Lazy Adam / RMSprop: replay every lagging row up to the optimizer clock, so that the tables (and the
        moments) hold what the dense sweep would have left.  Called before anything reads them; cheap when nothing
        lags."""
        if self._lazy is not None and self._lazy["dirty"]:
            self.k.lazy_flush(self._lazy)
            self._lazy["dirty"] = False

    def _dense_opt_step(self):
        """One dense sweep (the per-step paths); with lazy state around: flush first, then every row is current."""
        self.flush_lazy()
        self.k.opt_step(self.optimizer, self.model.flat, self._g_flat, self.step_count)
        if self._lazy is not None:
            self.k.lazy_mark_current(self._lazy)

    def gather_full_state_dict(self):
        """All-gather the shards into a reference-compatible state_dict (every rank gets it; rank 0
        typically ``torch.save``s it — the checkpoint format of torch_engine.py:70-73)."""
        self.flush_lazy()
        R = self.world
        full = {}
        for k, v in self.model.state_dict().items():
            if k == "global_bias":
                full[k] = v.detach().clone()
                continue
            n_total = self.n_users if k.startswith("user") else self.n_items
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

    # ---- checkpoints ------------------------------------------------------------------------
    OPT_STATE_FORMAT = "hiprec-sharded-optimizer-state-1"

    def save_checkpoint(self, model_dir, optimizer_state=False):
        """Collective.  Rank 0 writes ONE reference-format file (torch_engine.py:70-73: the gathered state_dict, as a
        single-GPU engine would) to ``model_dir``.  ``optimizer_state=True``: every rank also writes its shard's
        optimizer state -- the clock as the device holds it, the moments, the lazy form's stamps and scalars after a
        flush -- to ``model_dir + ".opt.rank<r>of<R>"`` (resumable at the same world size)."""
        full = self.gather_full_state_dict()          # flushes the lazy rows first
        if self.rank == 0:
            # written beside its final name and renamed: no rank can ever open a half-written file
            tmp = f"{model_dir}.tmp{os.getpid()}"
            torch.save({k: v.cpu() for k, v in full.items()}, tmp)
            os.replace(tmp, model_dir)
        # ... and nobody returns before it is there: an immediate resume_checkpoint runs torch.load on every rank
        token = torch.zeros(1, dtype=torch.int32, device=self.device if dist.get_backend(self.pg) == "nccl" else "cpu")
        dist.all_reduce(token, group=self.pg)
        token.cpu()                                   # the host waits: rank 0 reduces only after its rename
        if optimizer_state:
            if not isinstance(self.k, HipKernels):
                raise NotImplementedError("optimizer checkpoints need the HIP kernels' device state")
            opt, lz = self.optimizer, self._lazy
            cpu = lambda t: None if t is None else t.detach().to("cpu", copy=True)   # noqa: E731
            torch.save({"format": self.OPT_STATE_FORMAT, "optimizer": opt.name, "world": self.world, "rank": self.rank,
                        "n_params": int(self.model.flat.numel()), "step_count": int(self.step_count),
                        "hyper": {"lr": opt.lr, "beta1": opt.beta1, "beta2": opt.beta2, "eps": opt.eps},
                        "stats": cpu(self.k.stats), "exp_avg": cpu(opt.exp_avg), "exp_avg_sq": cpu(opt.exp_avg_sq),
                        "lazy": None if lz is None else {k: cpu(lz[k]) for k in ("stamp_u", "stamp_i", "scalars")}},
                       f"{model_dir}.opt.rank{self.rank}of{self.world}")

    def resume_checkpoint(self, model_dir, optimizer_state=False):
        """Every rank keeps its rows of the reference-format file; ``optimizer_state=True`` also restores this rank's
        ``.opt.rank<r>of<R>`` file."""
        self.load_full_state_dict(torch.load(model_dir, map_location="cpu"))
        if not optimizer_state:
            return self.model
        payload = torch.load(f"{model_dir}.opt.rank{self.rank}of{self.world}", map_location="cpu")
        opt = self.optimizer
        if (payload.get("format") != self.OPT_STATE_FORMAT or payload["optimizer"] != opt.name
                or payload["world"] != self.world or payload["n_params"] != self.model.flat.numel()):
            raise ValueError("the optimizer state file does not fit this engine (format / optimizer / world size / shard size)")
        # the raw stats block carries the running beta powers and the lazy scalars table the per-step learning rates: a
        # file written under other hyper-parameters would continue THAT run under this engine's name
        hyper = payload.get("hyper")
        mine = {"lr": opt.lr, "beta1": opt.beta1, "beta2": opt.beta2, "eps": opt.eps}
        if hyper is not None and any(float(hyper[k]) != float(mine[k]) for k in mine):
            raise ValueError(f"the optimizer state was saved with {hyper}, this engine runs {mine}")
        dev = self.device
        self.k.stats.copy_(payload["stats"].to(dev))
        self.step_count = int(payload["step_count"])
        for attr in ("exp_avg", "exp_avg_sq"):
            if getattr(opt, attr) is not None:
                getattr(opt, attr).copy_(payload[attr].to(dev))
        if self._lazy is not None:
            if payload["lazy"] is not None:
                for k in ("stamp_u", "stamp_i", "scalars"):
                    self._lazy[k].copy_(payload["lazy"][k].to(dev))
            else:
                self.k.lazy_mark_current(self._lazy)
            self._lazy["dirty"] = False
        return self.model

    # ---- exchange helpers -------------------------------------------------------------------
    def _exchange_counts(self, counts):
        """counts[q] = how many items I send to rank q  ->  how many I receive from each rank."""
        recv = torch.empty_like(counts)
        dist.all_to_all_single(recv, counts, group=self.pg)
        return recv

    def _a2a(self, send, send_counts, recv_counts, group=None):
        """Variable-size all-to-all of rows (first dim split by the per-rank counts)."""
        out = torch.empty((int(sum(recv_counts)),) + tuple(send.shape[1:]), dtype=send.dtype,
                          device=send.device)
        dist.all_to_all_single(out, send.contiguous(), output_split_sizes=list(recv_counts),
                               input_split_sizes=list(send_counts), group=group or self.pg)
        return out

    def _bucket(self, owner):
        """Stable grouping by destination rank: (order, send_counts host list)."""
        order = torch.argsort(owner, stable=True)
        counts = torch.bincount(owner, minlength=self.world)
        return order, counts

    # ---- one step ---------------------------------------------------------------------------
    def train_single_batch(self, batch_data, sync=True):
        """One optimisation step on the GLOBAL batch formed by every rank's ``batch_data``.
        Returns ``(loss, regularizer)`` of the global batch (identical on every rank)."""
        if self.routing == "padded":
            return self._step_padded(batch_data, sync)
        return self._step_variable(batch_data, sync)

    def _padded_buffers(self, b):
        """Persistent device buffers of the padded step for local batch size b (allocated once)."""
        key = (b, self.world)
        pb = getattr(self, "_pb", None)
        if pb is not None and pb["key"] == key:
            return pb
        R, dev, D = self.world, self.device, self.emb_dim
        cap1 = int(b / R * self.route_slack) + 64
        cap2 = int(2 * b / R * self.route_slack + 0.15 * b) + 64
        T1, T2 = R * cap1, R * cap2
        i64 = dict(dtype=torch.int64, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        pb = {"key": key, "cap1": cap1, "cap2": cap2, "T1": T1, "T2": T2,
              "send": torch.empty((T1, 3), **i64), "recv": torch.empty((T1, 3), **i64),
              "req": torch.empty(T2, **i64), "incoming": torch.empty(T2, **i64),
              "slot_pos": torch.empty(T1, **i64), "slot_neg": torch.empty(T1, **i64),
              "u_loc": torch.empty(T1, **i64), "local_idx": torch.empty(T2, **i64),
              "payload": torch.empty((T2, D + 1), **f32), "fetched": torch.empty((T2, D + 1), **f32),
              "rows": torch.empty((T2, D), **f32), "bias": torch.empty((T2, 1), **f32),
              # gradient of the fetched rows: rows and biases back to back so that one fill clears both
              "g_flat": torch.empty(T2 * (D + 1), **f32),
              "g_send": torch.empty((T2, D + 1), **f32), "g_recv": torch.empty((T2, D + 1), **f32),
              "part": torch.zeros(3, **f32)}
        pb["g_rows"] = pb["g_flat"][: T2 * D].view(T2, D)
        pb["g_bias"] = pb["g_flat"][T2 * D:].view(T2, 1)
        self._pb = pb
        return pb

    def _step_padded(self, batch_data, sync=True):
        """Fixed-capacity routing: every exchange moves world x cap slots (padding = -1), bucketing and
        packing run on the device in one launch each (csrc/shard.hip), the buffers are persistent, and
        nothing is read back by the host during the step.  Every rank must pass the same local batch
        size."""
        R, dev, m, D = self.world, self.device, self.model, self.emb_dim
        users, pos, neg = (torch.as_tensor(x, device=dev).to(torch.int64).contiguous()
                           for x in batch_data)
        b = users.numel()
        B = b * R
        if B < 2:
            raise IndexError("Dimension out of range (expected to be in range of [-1, 0], but got 1)")
        pb = self._padded_buffers(b)
        k = self.k

        # A2A-1: triples to the owner of the user row
        k.route_triples(users, pos, neg, R, pb["cap1"], pb["send"])
        dist.all_to_all_single(pb["recv"], pb["send"], group=self.pg)

        # A2A-2: item ids to their owners, rows (+bias) back into the SAME slots
        k.route_items(pb["recv"], R, pb["cap2"], pb["req"], pb["slot_pos"], pb["slot_neg"], pb["u_loc"])
        dist.all_to_all_single(pb["incoming"], pb["req"], group=self.pg)
        k.gather_payload(m.item_emb.weight.data, m.item_bias.weight.data, pb["incoming"], R, pb["payload"],
                         pb["local_idx"])
        dist.all_to_all_single(pb["fetched"], pb["payload"], group=self.pg)
        k.split_rows(pb["fetched"], pb["rows"], pb["bias"])

        # the single-GPU gradient kernel: the fetched buffer plays the item table, slot ids are the
        # item indices, padded triples (user -1) are skipped by the kernel
        gue, gie, gub, gib, ggb = m._views(self._g_flat)
        w = {"user_emb.weight": m.user_emb.weight.data, "user_bias.weight": m.user_bias.weight.data,
             "global_bias": m.global_bias.data, "item_emb.weight": pb["rows"], "item_bias.weight": pb["bias"]}
        pb["g_flat"].zero_()
        g = {"user_emb.weight": gue, "user_bias.weight": gub, "global_bias": ggb,
             "item_emb.weight": pb["g_rows"], "item_bias.weight": pb["g_bias"]}
        part = pb["part"]
        k.bpr_grad_into(w, g, pb["u_loc"], pb["slot_pos"], pb["slot_neg"], 1.0 / B, float(self.reg), part)

        # A2A-3: item-row gradients back to the owners (same slots, reverse direction)
        k.join_rows(pb["g_rows"], pb["g_bias"], pb["g_send"])
        dist.all_to_all_single(pb["g_recv"], pb["g_send"], group=self.pg)
        k.scatter_add_rows(gie, pb["local_idx"], pb["g_recv"][:, :D])
        k.scatter_add_rows(gib, pb["local_idx"], pb["g_recv"][:, D:])

        dist.all_reduce(part, group=self.pg)
        ggb += part[2]
        self.step_count += 1
        if self._rows_sgd:
            # the step touched the user rows it received and the item rows its peers asked for
            if self._stamp > 2**31 - 8:
                self._user_stamp.zero_()
                self._item_stamp.zero_()
                self._stamp = 0
            k.sgd_rows(m, self._g_flat, pb["u_loc"], None, self.optimizer.lr, self._user_stamp,
                       self._item_stamp, self._stamp + 1)
            k.sgd_rows(m, self._g_flat, None, pb["local_idx"], self.optimizer.lr, self._user_stamp,
                       self._item_stamp, self._stamp + 2)
            self._stamp += 2
        else:
            self._dense_opt_step()
        if not sync:
            self._pending = part
            return None
        k.check_status()
        self.last = (float(part[0]), float(part[1]))
        return self.last

    def _step_variable(self, batch_data, sync=True):
        """Exact-size routing with host-side split sizes (any per-rank batch sizes)."""
        R = self.world
        dev = self.device
        users, pos, neg = (torch.as_tensor(x, device=dev).to(torch.int64).contiguous()
                           for x in batch_data)
        b_local = torch.tensor([users.numel()], dtype=torch.int64, device=dev)
        dist.all_reduce(b_local, group=self.pg)
        B = int(b_local.item())
        if B < 2:
            raise IndexError("Dimension out of range (expected to be in range of [-1, 0], but got 1)")

        # A2A-1: triples to the rank that owns the user row
        order, counts = self._bucket(users % R)
        send = torch.stack([users, pos, neg], dim=1)[order]
        recv_counts = self._exchange_counts(counts)
        sc, rc = counts.tolist(), recv_counts.tolist()
        mine = self._a2a(send, sc, rc)
        u_g, p_g, n_g = mine[:, 0].contiguous(), mine[:, 1].contiguous(), mine[:, 2].contiguous()
        b_r = u_g.numel()

        # A2A-2: fetch the item rows (slot k = pos of triple k, slot b_r + k = its neg)
        items = torch.cat([p_g, n_g])
        iorder, icounts = self._bucket(items % R)
        req = items[iorder]
        irecv_counts = self._exchange_counts(icounts)
        isc, irc = icounts.tolist(), irecv_counts.tolist()
        incoming = self._a2a(req, isc, irc)          # ids of MY rows that peers ask for
        m = self.model
        local_idx = torch.div(incoming, R, rounding_mode="floor")
        D = self.emb_dim
        payload = torch.cat([self.k.gather_rows(m.item_emb.weight.data, local_idx),
                             self.k.gather_rows(m.item_bias.weight.data, local_idx)], dim=1)
        fetched_sorted = self._a2a(payload, irc, isc)   # rows come back in request order
        fetched = torch.empty_like(fetched_sorted)
        fetched[iorder] = fetched_sorted

        # the single-GPU gradient kernel on (local user shard, fetched item rows)
        w = {"user_emb.weight": m.user_emb.weight.data, "user_bias.weight": m.user_bias.weight.data,
             "global_bias": m.global_bias.data,
             "item_emb.weight": fetched[:, :D].contiguous(),
             "item_bias.weight": fetched[:, D:].contiguous()}
        gue, gie, gub, gib, ggb = m._views(self._g_flat)
        g_rows = torch.zeros((2 * b_r, D), dtype=torch.float32, device=dev)
        g_bias = torch.zeros((2 * b_r, 1), dtype=torch.float32, device=dev)
        g = {"user_emb.weight": gue, "user_bias.weight": gub, "global_bias": ggb,
             "item_emb.weight": g_rows, "item_bias.weight": g_bias}
        slots = torch.arange(2 * b_r, dtype=torch.int64, device=dev)
        u_loc = torch.div(u_g, R, rounding_mode="floor")
        if b_r > 0:
            part = self.k.bpr_grad(w, g, u_loc, slots[:b_r].contiguous(), slots[b_r:].contiguous(),
                                   1.0 / B, float(self.reg))
        else:
            part = torch.zeros(3, dtype=torch.float32, device=dev)
            self.k.advance_clock()

        # A2A-3: item-row gradients back to their owners, scatter-add into the local dense gradient
        gpayload = torch.cat([g_rows, g_bias], dim=1)[iorder]
        gincoming = self._a2a(gpayload, isc, irc)
        if gincoming.shape[0] > 0:
            self.k.scatter_add_rows(gie, local_idx, gincoming[:, :D])
            self.k.scatter_add_rows(gib, local_idx, gincoming[:, D:])

        # loss, regularizer and d loss / d global_bias are sums over all ranks
        dist.all_reduce(part, group=self.pg)
        ggb += part[2]
        self.step_count += 1
        self._dense_opt_step()
        if not sync:
            self._pending = part
            return None
        self.k.check_status()
        loss, reg = float(part[0]), float(part[1])
        self.last = (loss, reg)
        return loss, reg

    # ---- epoch-planned steps: routing is a property of the DATA, so it is done once per epoch ---------------------------
    # Which rank owns a triple's user row and which item rows each rank must fetch for each step depend on the ids
    # only, not on the weights.  For a device-resident loader the whole epoch is therefore routed in the staging:
    # ONE all-to-all moves every triple to owner(user); the item references of every step are de-duplicated per
    # (step, owner) (Zipf items: high duplication) and ONE all-to-all tells every owner which rows it will
    # be asked for, step by step.  What is left per step is exact-size (the split sizes are host integers: no
    # padding, nothing read back) and weight-dependent only:
    #     payload rows (+ clear) -> exchange -> gradient kernel on (local users, fetched rows) -> exchange -> apply
    # i.e. 2 exchanges and 4 launches.  Plain SGD: user rows are updated in place by the gradient kernel
    # (csrc/mf_owned.hip), item gradients are summed per fetched slot and applied by the owner with -lr straight
    # into the table: no dense gradient buffer, no touched-rows pass.  Adam / RMSprop: the same launches accumulate
    # into the shard's dense gradient and one local sweep follows (their moments move every element every step,
    # torch_engine.py:30-39).  Loss / reg / scalar-bias partials ride in one extra row per peer of the gradient
    # exchange instead of a separate all-reduce.  The planner itself is csrc/plan.hip (round 2: torch sorts).
    def plan_epoch(self, train_loader, group=None):
        """Collective.  Route one epoch of a DeviceTripleBatcher-like loader (this rank's share; the same number of
        triples and batch size on every rank).  Returns the plan :meth:`run_planned_epoch` consumes.  group: the
        process group its exchanges use (default: the engine's; :meth:`prefetch_plan` passes one of its own)."""
        pg = group or self.pg
        R, dev, k = self.world, self.device, self.k
        if self.emb_dim < 2:
            raise ValueError("the planned sharded epoch needs emb_dim >= 2 (an exchange row carries three partial sums)")
        if max(self.n_users, self.n_items) >= 2**31:
            raise ValueError("the epoch planner works on 32-bit ids: n_users, n_items < 2^31")
        users, pos, neg = (getattr(train_loader, a).to(dev, torch.int64).contiguous()
                           for a in ("user_tensor", "pos_item_tensor", "neg_item_tensor"))
        n, bs = users.numel(), int(train_loader.batch_size)
        sizes = torch.tensor([n, -n, bs, -bs], dtype=torch.int64, device=dev)
        dist.all_reduce(sizes, op=dist.ReduceOp.MAX, group=pg)
        if sizes[0] != -sizes[1] or sizes[2] != -sizes[3]:
            raise ValueError("the planned sharded epoch needs the same number of triples and batch size on every rank")
        if n == 0:
            raise ValueError("the planned sharded epoch needs a non-empty loader")
        S = (n + bs - 1) // bs
        perm = train_loader.permutation()
        if perm is not None:
            perm = perm.to(dev, torch.int64).contiguous()

        # (1) triples -> owner(user), the whole epoch in one exchange, (destination, step)-ordered
        send, cnt_ds = k.plan_route(users, pos, neg, perm, bs, R, self.n_users, self.n_items)
        recv_cnt = torch.empty_like(cnt_ds)
        dist.all_to_all_single(recv_cnt, cnt_ds, group=pg)                 # [source, step]
        # one rank's out-of-range id must stop EVERY rank here, between the same two collectives (raised
        # by its holder only, the peers walked on into the next all-to-all and hung): the status words are OR-ed
        # (bit masks: MAX would lose bits) over the group before the host reads them
        status = k.plan_status().to(cnt_ds.device)
        if R > 1:
            bits = torch.stack([(status >> b) & 1 for b in range(8)]).reshape(-1)
            dist.all_reduce(bits, op=dist.ReduceOp.MAX, group=pg)
            status = (bits << torch.arange(8, device=bits.device)).sum().reshape(1)
        host = torch.cat([cnt_ds.sum(1), recv_cnt.sum(1), recv_cnt.sum(0).max().reshape(1), status]).tolist()  # host sync 1 of 2
        status = int(host[2 * R + 1])
        if status:
            k.clear_status()
            raise_on_status(status)   # IndexError, as nn.Embedding raises; the status word is cleared, the engine usable
        send1, recv1, cap = host[:R], host[R:2 * R], max(int(host[2 * R]), 1)
        recv = self._a2a(send[:sum(send1)], send1, recv1, pg)               # (source, step)-ordered
        U, P, N, fill = k.plan_place_triples(recv, recv_cnt, S, cap)        # fixed-size blocks per step, user -1 = padding

        # (2) the step's distinct items become slots of its exchange buffer ([rows asked of q ..., 1 extra row] per
        # owner q), the blocks are re-laid grouped by positive item with their row-ownership arrays
        sl = k.plan_item_slots(U, P, N, S, cap, R, self.model.n_users, self.n_items, fill)

        # (2b) plain SGD, C step driver: the step blocks' contribution lists -- the step then runs as owner pulls
        # (hiprec_mf_bpr_pull_remote_step: no float atomics, nothing to clear, the partials' publish rides along)
        contrib = None
        if self._pull_steps():
            from .mf import batch_row_contributions

            contrib = batch_row_contributions(sl["U"], sl["SP"], sl["SN"], cap, max(self.model.n_users, 1),
                                              sl["slot_bound"], self.emb_dim)

        # (3) tell every owner which rows it will be asked for, step by step: one exchange
        req_ds = sl["req_ds"]
        in_qs = torch.empty_like(req_ds)
        dist.all_to_all_single(in_qs, req_ds, group=pg)                     # [source, step]
        host = torch.cat([req_ds.sum(1), in_qs.sum(1), sl["req_cnt"].reshape(-1),
                          in_qs.t().reshape(-1)]).tolist()                 # host sync 2 of 2
        send2, recv2 = host[:R], host[R:2 * R]
        req_l = [host[2 * R + j * R: 2 * R + (j + 1) * R] for j in range(S)]
        in_l = [host[2 * R + S * R + j * R: 2 * R + S * R + (j + 1) * R] for j in range(S)]
        incoming = self._a2a(sl["req_send"][:sum(send2)], send2, recv2, pg)  # (source, step)-ordered
        placed = k.plan_place_requests(incoming, in_qs, S, self.model.n_items)
        in_idx, ex_in, dup_bits = placed if len(placed) == 3 else (placed + (None,))
        in_len = [sum(c) + R for c in in_l]
        n_slots = [sum(c) + R for c in req_l]
        in_off = [0]
        for length in in_len:
            in_off.append(in_off[-1] + length)
        return {"S": S, "cap": cap, "bs": bs, "n": n, "world": R, "rank": self.rank, "U": sl["U"], "SP": sl["SP"],
                "SN": sl["SN"], "own": sl["own"], "total": sl["total"], "stride": sl["stride"], "in_idx": in_idx,
                "in_off": in_off[:-1], "in_len": in_len, "n_slots": n_slots,
                "req_split": [[c + 1 for c in row] for row in req_l], "in_split": [[c + 1 for c in row] for row in in_l],
                "ex_req": sl["ex_req"], "ex_in": ex_in, "slot_shared": sl.get("slot_shared"), "dup_bits": dup_bits,
                "in_off_h": in_off, "n_slots_h": n_slots, "req_cnt_h": req_l, "in_cnt_h": in_l, "contrib": contrib}

    def _pull_steps(self):
        """``shard_sgd``: "pull" (default) -- a planned epoch's plain-SGD steps run as owner pulls when the C step
        driver enqueues them and emb_dim % 4 == 0 (two launches without float atomics, csrc/mf_owned.hip); "atomic" --
        the round 2-4 step (shared rows and shared slots through device-scope atomics).  The torch step loop and the
        dense optimizers keep the atomic kernel."""
        mode = self.config["model"].get("shard_sgd", "pull")
        if mode not in ("pull", "atomic"):
            raise ValueError(f"shard_sgd must be 'pull' or 'atomic', not {mode!r}")
        return (mode == "pull" and self.optimizer.name == "sgd" and self.emb_dim % 4 == 0 and self.emb_dim >= 4
                and self._step_comm() == "c")

    def prefetch_setup(self):
        """Collective, one-off: the side stream and the process group (a communicator of its own: no ordering against
        the steps' exchanges) the prefetched plans use.  :meth:`prefetch_plan` calls it; callers that time epochs
        call it beforehand (creating a communicator takes tens of milliseconds)."""
        if getattr(self, "_plan_stream", None) is None:
            self._plan_stream = torch.cuda.Stream(device=self.device)
            ranks = None if self.pg is None else dist.get_process_group_ranks(self.pg)
            self._plan_pg = dist.new_group(ranks=ranks, like=self.pg)
            if dist.get_backend(self._plan_pg) == "nccl":   # RCCL creates the communicator lazily: do it now
                with torch.cuda.stream(self._plan_stream):
                    t = torch.zeros(self.world, dtype=torch.int32, device=self.device)
                    dist.all_to_all_single(torch.empty_like(t), t, group=self._plan_pg)
                self._plan_stream.synchronize()

    def prefetch_plan(self, train_loader):
        """Collective.  Plan the NEXT epoch now, on a side stream and over a process group of its own, while the
        steps of the current one (already enqueued) run: the plan depends on the data only -- its sorts, its four
        exchanges and its two host round trips (4.9 ms per 30 steps of 65 536 triples) then cost the training
        stream nothing.  :meth:`take_plan` hands it to the next epoch."""
        self.prefetch_setup()
        begin = getattr(self, "_ev_epoch_begin", None)
        if begin is not None:
            # the plan of epoch k + 2 is made while epoch k + 1 runs, not earlier: the host cannot run epochs ahead
            # of the GPU (the plan's host reads wait here), and at most two plans are in flight
            self._plan_stream.wait_event(begin)
        with torch.cuda.stream(self._plan_stream):
            plan = self.plan_epoch(train_loader, group=self._plan_pg)
            plan["ready"] = torch.cuda.Event()
            plan["ready"].record(self._plan_stream)
        self._prefetched_plan = plan
        return plan

    def take_plan(self, train_loader):
        """The plan of the epoch about to run: the prefetched one (the training stream waits for its event) or a
        fresh synchronous one."""
        plan = getattr(self, "_prefetched_plan", None)
        self._prefetched_plan = None
        main = torch.cuda.current_stream(self.device)
        if plan is None:
            plan = self.plan_epoch(train_loader)
        else:
            main.wait_event(plan["ready"])
            # the plan's tensors come from the side stream's allocator pool and are read by the training stream:
            # the allocator must not hand their memory to a later plan before the steps that read them are done
            # (r02 experiments 24; with the C step driver the host is far ahead of the GPU)
            for v in list(plan.values()) + list(plan.get("contrib") or ()):
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(main)
        self._ev_epoch_begin = torch.cuda.Event()
        self._ev_epoch_begin.record(main)
        return plan

    def _step_comm(self):
        """How the planned steps are enqueued: "c" -- one C call per range of steps that launches the kernels AND
        posts the exchanges (grouped ncclSend / ncclRecv on a communicator of the engine's own, _rccl.py; no
        communicator at all at world size 1) -- or "torch" (torch.distributed.all_to_all_single between the launches:
        the gloo tests' path, and the fallback when the RCCL binding is not available on every rank).
        `step_driver: "torch"` in the model config forces the latter."""
        mode = getattr(self, "_step_mode", None)
        if mode is not None:
            return mode
        self._comm = None
        want_c = isinstance(self.k, HipKernels) and self.config["model"].get("step_driver", "c") == "c"
        # `shard_self_exchange: True`: every rank also sends its own segment of the two exchanges to itself through
        # the communicator (bit-identical; at world size 1 it is what executes the real ncclSend / ncclRecv path)
        self._self_exchange = bool(self.config["model"].get("shard_self_exchange", False))
        if self._self_exchange and not want_c:
            raise ValueError("shard_self_exchange is an option of the C step driver (step_driver: 'c')")
        if want_c and (self.world > 1 or self._self_exchange):
            from . import _rccl

            comm = _rccl.create_communicator(self.pg, self.device)
            ok = comm is not None and comm.has_send_recv()
            if _rccl.all_ranks_agree(ok, self.pg, self.device):
                self._comm = comm
                self.k.comm_watch = (comm, float(self.config["model"].get("collective_timeout_s", 300.0)))
            else:
                want_c = False
                if comm is not None:
                    comm.destroy()
                if self._self_exchange:
                    raise RuntimeError("shard_self_exchange: no RCCL communicator with ncclSend / ncclRecv could be made")
        self._step_mode = "c" if want_c else "torch"
        return self._step_mode

    def run_planned_epoch(self, plan, steps=None, sync=True):
        """Collective.  Enqueue every step of a planned epoch, or steps [a, b) of it; nothing is read back until the
        end.  Returns (last loss, last reg, loss sum, reg sum) of the global batches (None with ``sync=False``)."""
        R, dev, m, D, k = self.world, self.device, self.model, self.emb_dim, self.k
        ld, cap, S = D + 1, plan["cap"], plan["S"]
        dense = self.optimizer.name != "sgd"
        max_in, max_slots = max(plan["in_len"]), max(plan["n_slots"])
        pb = getattr(self, "_planned_bufs", None)
        if pb is None or pb["max_in"] < max_in or pb["max_slots"] < max_slots or pb["stride"] < plan["stride"]:
            f32 = dict(dtype=torch.float32, device=dev)
            pb = self._planned_bufs = {
                "max_in": max_in, "max_slots": max_slots, "stride": plan["stride"],
                "payload": torch.empty((max_in, ld), **f32), "g_recv": torch.empty((max_in, ld), **f32),
                "fetched": torch.empty((max_slots, ld), **f32), "g_send": torch.empty((max_slots, ld), **f32),
                "arrived": torch.zeros(plan["stride"], dtype=torch.int32, device=dev),
                "acc": torch.zeros(plan["stride"] * ld, **f32)}
        if plan.get("contrib") is not None and (pb.get("cbuf") is None or pb["cbuf"].numel() < 3 * cap * D):
            pb["cbuf"] = torch.empty(3 * cap * D, dtype=torch.float32, device=dev)    # (work space: never cleared)
            pb["cbias"] = torch.empty(3 * cap, dtype=torch.float32, device=dev)
        lr, reg = self.optimizer.lr, float(self.reg)
        a, b = steps or (0, S)
        lazy = self._lazy if dense else None
        if lazy is not None:
            lazy["dirty"] = True
        if self._step_comm() == "c":
            k.planned_steps(plan, pb, m, self._g_flat, self.optimizer, a, b, reg, self._comm, lazy,
                            self_exchange=self._self_exchange)
            self.step_count += b - a
            if b == S and self.flush_lazy_every_epoch:
                self.flush_lazy()     # the epoch's callers (evaluation, checkpoints) read the tables
            return k.epoch_stats() if sync else None
        item_emb, item_bias = m.item_emb.weight.data, m.item_bias.weight.data
        gue, gie, gub, gib, ggb = m._views(self._g_flat)
        for s in range(a, b):
            il, sl = plan["in_len"][s], plan["n_slots"][s]
            idx = plan["in_idx"][plan["in_off"][s]: plan["in_off"][s] + il]
            payload, fetched = pb["payload"][:il], pb["fetched"][:sl]
            g_send, g_recv = pb["g_send"][:sl], pb["g_recv"][:il]
            blk = slice(s * cap, (s + 1) * cap)
            if lazy is not None:
                k.lazy_catchup(lazy, plan["U"][blk], idx)
            k.payload_zero(item_emb, item_bias, idx, payload, g_send)
            dist.all_to_all_single(fetched, payload, output_split_sizes=plan["req_split"][s],
                                   input_split_sizes=plan["in_split"][s], group=self.pg)
            blk = slice(s * cap, (s + 1) * cap)
            B = R * min(plan["bs"], plan["n"] - s * plan["bs"])
            if dense:
                k.grad_remote_step(m, self._g_flat, fetched, g_send, sl, plan["U"][blk], plan["SP"][blk],
                                   plan["SN"][blk], plan["own"][:, blk], plan["total"][s], 1.0 / B, reg)
            else:
                k.owned_remote_step(m, fetched, g_send, sl, plan["U"][blk], plan["SP"][blk], plan["SN"][blk],
                                    plan["own"][:, blk], plan["total"][s], pb["arrived"], pb["acc"], 1.0 / B, reg, lr)
            k.publish_partials(g_send, D, plan["ex_req"][s])
            dist.all_to_all_single(g_recv, g_send, output_split_sizes=plan["in_split"][s],
                                   input_split_sizes=plan["req_split"][s], group=self.pg)
            self.step_count += 1
            if dense:
                k.apply_finish(gie, gib, idx, g_recv, 1.0, plan["ex_in"][s], ggb, 1.0, s == 0)
                if lazy is not None:
                    k.lazy_update(lazy, plan["U"][blk], idx)
                else:
                    k.opt_step(self.optimizer, m.flat, self._g_flat, self.step_count)
            else:
                k.apply_finish(item_emb, item_bias, idx, g_recv, -lr, plan["ex_in"][s], m.global_bias.data, -lr, s == 0)
        if b == S and self.flush_lazy_every_epoch:
            self.flush_lazy()
        return k.epoch_stats() if sync else None

    def _equal_loaders(self, train_loader):
        """Collective.  The planned epoch needs the same number of triples and batch size on every rank; loaders that
        differ take the per-batch loop (fall back instead of raising)."""
        n, bs = len(train_loader.user_tensor), int(train_loader.batch_size)
        t = torch.tensor([n, -n, bs, -bs], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
        return n > 0 and self.emb_dim >= 2 and int(t[0]) == -int(t[1]) and int(t[2]) == -int(t[3])

    def train_an_epoch(self, train_loader, epoch_id):
        """Every rank iterates its own shard of the interaction stream; all loaders must yield the
        same number of batches (one collective step per batch).  A device-resident loader takes the epoch-planned
        path (plan_epoch + run_planned_epoch), whatever the optimizer."""
        if (all(hasattr(train_loader, a) for a in ("user_tensor", "pos_item_tensor", "neg_item_tensor", "permutation"))
                and self.config["model"].get("epoch_plan", True) and self._equal_loaders(train_loader)):
            loss, _, total_loss, total_reg = self.run_planned_epoch(self.plan_epoch(train_loader))
            if self.rank == 0:
                print(f"[Training Epoch {epoch_id}], Loss {loss}, Regularizer {total_reg}")
            return total_loss, total_reg
        total_loss, total_reg, loss = 0.0, 0.0, float("nan")
        for batch in train_loader:
            loss, reg = self.train_single_batch(batch)
            total_loss += loss
            total_reg += reg
        if self.rank == 0:
            print(f"[Training Epoch {epoch_id}], Loss {loss}, Regularizer {total_reg}")
        return total_loss, total_reg
