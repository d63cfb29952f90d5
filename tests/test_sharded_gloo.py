"""World-size-2 test of the row-sharded engine's HOST logic on CPU (gloo): triple routing, item-row
fetch, gradient return, all-reduce, shard bookkeeping.  The HIP kernels cannot run here, so the
oracle stands in for them through the engine's kernel-backend seam (tests may use the oracle; the
product default is HipKernels, which refuses to run without a GPU)."""
import contextlib
import io
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import KEYS, assert_scalar_close, assert_tensor_close
from oracle import mf_numpy as onp
import plan_statements as ps


class OracleKernels:
    """numpy-oracle implementation of the sharded engine's kernel backend (CPU tensors)."""

    def __init__(self):
        self.st = None

    def reset_clock(self, beta1, beta2):
        self.clock = 0

    def gather_rows(self, table, idx):
        out = table[idx.clamp(min=0)].clone()
        out[idx < 0] = 0.0  # padding slots yield zero rows
        return out

    def route_bucket(self, keys, n_dest, cap):
        slots = torch.full_like(keys, -1)
        fill = [0] * n_dest
        for k, key in enumerate(keys.tolist()):
            if key < 0:
                continue
            d = key % n_dest
            if fill[d] < cap:
                slots[k] = d * cap + fill[d]
                fill[d] += 1
            else:
                self.overflow = True
        return slots

    # ---- the fused routing / packing ops of the padded step (csrc/shard.hip), restated with numpy ------
    def route_triples(self, users, pos, neg, n_dest, cap, send):
        send.fill_(-1)
        slots = self.route_bucket(users, n_dest, cap)
        for k, s in enumerate(slots.tolist()):
            if s >= 0:
                send[s, 0], send[s, 1], send[s, 2] = users[k], pos[k], neg[k]

    def route_items(self, recv, n_dest, cap, req, slot_pos, slot_neg, u_loc):
        req.fill_(-1)
        u, p, n = recv[:, 0], recv[:, 1].clone(), recv[:, 2].clone()
        p[u < 0] = -1
        n[u < 0] = -1
        # the kernel buckets each triple's pos then neg, wave by wave; any arrival order is legal
        slots = self.route_bucket(torch.stack([p, n], 1).reshape(-1), n_dest, cap).reshape(-1, 2)
        for k in range(recv.shape[0]):
            sp, sn = int(slots[k, 0]), int(slots[k, 1])
            if sp >= 0:
                req[sp] = p[k]
            if sn >= 0:
                req[sn] = n[k]
            live = u[k] >= 0 and sp >= 0 and sn >= 0
            slot_pos[k], slot_neg[k] = (sp, sn) if live else (0, 0)
            u_loc[k] = u[k] // n_dest if live else -1

    def gather_payload(self, item_emb, item_bias, incoming, n_dest, payload, local_idx):
        local = torch.where(incoming < 0, incoming, torch.div(incoming, n_dest, rounding_mode="floor"))
        local_idx.copy_(local)
        payload.copy_(torch.cat([self.gather_rows(item_emb, local), self.gather_rows(item_bias, local)], dim=1))

    def split_rows(self, src, emb, bias):
        emb.copy_(src[:, :-1])
        bias.copy_(src[:, -1:])

    def join_rows(self, emb, bias, dst):
        dst.copy_(torch.cat([emb, bias], dim=1))

    def bpr_grad_into(self, w, g, users, pos, neg, inv_batch, reg_coef, part):
        part.copy_(self.bpr_grad(w, g, users, pos, neg, inv_batch, reg_coef))

    def sgd_rows(self, model, g_flat, users, items, lr, user_stamp, item_stamp, stamp):
        ue, ie, ub, ib, gb = model._views(model.flat)
        gue, gie, gub, gib, ggb = model._views(g_flat)
        none = torch.zeros(0, dtype=torch.int64)
        lr32 = torch.tensor(lr, dtype=torch.float32)
        for rows, w_e, w_b, g_e, g_b in (((users[users >= 0] if users is not None else none).unique(), ue, ub, gue, gub),
                                          ((items[items >= 0] if items is not None else none).unique(), ie, ib, gie, gib)):
            w_e[rows] = w_e[rows] - lr32 * g_e[rows]
            w_b[rows] = w_b[rows] - lr32 * g_b[rows]
            g_e[rows] = 0.0
            g_b[rows] = 0.0
        gb -= lr32 * ggb
        ggb.zero_()

    def scatter_add_rows(self, table, idx, src):
        keep = idx >= 0  # -1 = padding slot
        table.index_add_(0, idx[keep], src.contiguous()[keep])

    def bpr_grad(self, w, g, users, pos, neg, inv_batch, reg_coef):
        wn = {k: v.numpy() for k, v in w.items()}
        B = int(round(1.0 / inv_batch))
        live = users >= 0  # user -1 = padded triple slot
        users, pos, neg = users[live], pos[live], neg[live]
        self.clock += 1
        if users.numel() == 0:   # this rank received none of the step's triples (the kernel skips every slot)
            return torch.zeros(3, dtype=torch.float32)
        self.clock -= 1
        loss, reg, grads = onp.mf_bpr_grads(wn, users.numpy(), pos.numpy(), neg.numpy(), reg_coef,
                                            global_batch=B)
        for k in ("user_emb.weight", "item_emb.weight", "user_bias.weight", "item_bias.weight"):
            g[k] += torch.from_numpy(grads[k])
        self.clock += 1
        return torch.tensor([loss, reg, grads["global_bias"][0]], dtype=torch.float32)

    def advance_clock(self):
        self.clock += 1

    def opt_step(self, opt, flat_w, flat_g, step):
        if self.st is None:
            self.st = onp.new_opt_state({"flat": flat_w.numpy()}, opt.name)
        w = {"flat": flat_w.numpy()}  # shares memory with the tensor: in-place update
        onp.opt_step(w, {"flat": flat_g.numpy()}, self.st, opt.name, opt.lr)
        flat_g.zero_()

    def check_status(self):
        if getattr(self, "overflow", False):   # the kernels raise HIPREC_STATUS_ROUTE_OVERFLOW for this
            self.overflow = False
            raise RuntimeError("a fixed-capacity all-to-all bucket overflowed: raise the sharded engine's `route_slack`")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def make_config(U, I, D, optimizer, lr, routing="variable", sgd_mode="dense", **extra):
    return {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cpu", optimizer=optimizer,
                          lr=lr, batch_size=8, loss="bpr", routing=routing, sgd_mode=sgd_mode, **extra),
            "system": {"run_dir": "/tmp/hiprec_test_runs"}}


def worker(rank, world, port, optimizer, lr, splits, out_path, routing="variable", sgd_mode="dense"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import beta_recsys_amd  # noqa: F401
        from beta_recsys_amd.sharded import ShardedMFEngine

        U, I, D = 23, 19, 8
        w0 = onp.init_params(U, I, D, seed=7)
        rng = np.random.default_rng(100)
        with contextlib.redirect_stdout(io.StringIO()):
            eng = ShardedMFEngine(make_config(U, I, D, optimizer, lr, routing, sgd_mode), kernels=OracleKernels(),
                                  full_state={k: torch.from_numpy(v) for k, v in w0.items()})
        losses = []
        batches = []
        for split in splits:
            B = sum(split)
            users, pos, neg = rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)
            if world >= 8:          # Zipf positives: heavy duplication inside every owner's requests
                pz = 1.0 / np.arange(1, I + 1)
                pos = rng.choice(I, B, p=pz / pz.sum())
            pos[: B // 3] = pos[0]  # a popular item: many rows fetched from one owner
            batches.append((users, pos, neg))
            lo = sum(split[:rank])
            sl = slice(lo, lo + split[rank])
            losses.append(eng.train_single_batch((users[sl], pos[sl], neg[sl])))
        full = eng.gather_full_state_dict()
        if rank == 0:
            torch.save({"losses": losses, "full": {k: v.numpy() for k, v in full.items()},
                        "batches": batches, "w0": w0}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def check_steps(res, optimizer, lr):
    """Losses step by step and the gathered full state_dict against the single-process oracle on the concatenated
    batches: plain SGD every element within 1e-5 of the update, the dense optimizers every element inside the
    legal-trajectory envelope -- no allowance for a fraction of outliers."""
    from helpers import assert_on_trajectory, assert_sgd_exact, mf_trajectory

    w = onp.copy_params(res["w0"])
    st = onp.new_opt_state(w, optimizer)
    for (users, pos, neg), (loss, reg) in zip(res["batches"], res["losses"]):
        ref_loss, ref_reg = onp.mf_train_step(w, st, (users, pos, neg), "bpr", optimizer, lr)
        assert_scalar_close(loss, ref_loss, 2e-5, "loss")
        assert_scalar_close(reg, ref_reg, 2e-5, "reg")
    for k in KEYS:
        assert res["full"][k].shape == w[k].shape
    if optimizer == "sgd":
        assert_sgd_exact(res["full"], w, res["w0"], "sharded steps", lr=lr, batch=min(len(b[0]) for b in res["batches"]))
    else:
        w_ref, env, upd = mf_trajectory(res["w0"], res["batches"], optimizer, lr)
        assert_on_trajectory(res["full"], w_ref, env, upd, "sharded steps")


@pytest.mark.parametrize("optimizer,lr", [("sgd", 0.1), ("adam", 0.05), ("rmsprop", 0.01)])
def test_two_rank_sharded_step_equals_single_process(tmp_path, optimizer, lr):
    splits = [(10, 10), (13, 7), (20, 0), (1, 1)]  # even, uneven, one empty rank, tiny
    out_path = str(tmp_path / "out.pt")
    mp.spawn(worker, args=(2, free_port(), optimizer, lr, splits, out_path), nprocs=2, join=True)
    check_steps(torch.load(out_path, weights_only=False), optimizer, lr)


def test_eight_rank_variable_routing_equals_single_process(tmp_path):
    """Exact-size routing at world size 8 with Zipf items, uneven local batches, empty ranks, and 23 users / 19 items
    (n_rows % 8 != 0: shards of 3 and 2 rows)."""
    splits = [(5, 9, 0, 12, 7, 1, 0, 14), (6,) * 8, (0, 0, 0, 30, 0, 0, 0, 2)]
    out_path = str(tmp_path / "out.pt")
    mp.spawn(worker, args=(8, free_port(), "adam", 0.05, splits, out_path), nprocs=8, join=True)
    check_steps(torch.load(out_path, weights_only=False), "adam", 0.05)


@pytest.mark.parametrize("world,optimizer,lr", [(2, "sgd", 0.1), (2, "adam", 0.05), (4, "sgd", 0.1), (4, "adam", 0.05),
                                                (8, "sgd", 0.1)])
def test_two_rank_padded_routing_equals_single_process(tmp_path, optimizer, lr, world):
    """The fixed-capacity (no host sync) routing: same result as the single-process step -- on 2, 4 and 8 ranks
    (23 users and 19 items: n_rows % 4 != 0 and % 8 != 0, uneven shards; Zipf positives at world 8)."""
    splits = [(12,) * world, (30,) * world, (2,) * world]  # equal local batches, as the padded mode requires
    out_path = str(tmp_path / "out.pt")
    mp.spawn(worker, args=(world, free_port(), optimizer, lr, splits, out_path, "padded"), nprocs=world, join=True)
    check_steps(torch.load(out_path, weights_only=False), optimizer, lr)


def test_two_rank_padded_routing_with_touched_rows_sgd(tmp_path):
    """Plain SGD on shards visiting only the rows of the step (sgd_mode 'rows'): the same weights as the
    single-process dense step (untouched rows are not moved by SGD)."""
    splits = [(12, 12), (30, 30), (2, 2)]
    out_path = str(tmp_path / "out.pt")
    mp.spawn(worker, args=(2, free_port(), "sgd", 0.1, splits, out_path, "padded", "rows"), nprocs=2, join=True)
    check_steps(torch.load(out_path, weights_only=False), "sgd", 0.1)


def overflow_worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from beta_recsys_amd.sharded import ShardedMFEngine

        U, I, D, b = 64, 40, 4, 600
        with contextlib.redirect_stdout(io.StringIO()):
            eng = ShardedMFEngine(make_config(U, I, D, "sgd", 0.1, "padded"), kernels=OracleKernels())
        # every triple of every rank goes to the owner of user 0 and asks for item 0: no capacity built around the
        # mean (b / R * route_slack + headroom) holds that
        users, pos, neg = np.zeros(b, dtype=np.int64), np.zeros(b, dtype=np.int64), np.zeros(b, dtype=np.int64)
        with pytest.raises(RuntimeError, match="route_slack"):
            eng.train_single_batch((users, pos, neg))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_padded_routing_reports_a_full_bucket():
    """A bucket of the fixed-capacity exchange that overflows is an error (status bit -> RuntimeError naming
    `route_slack`), never a silent drop."""
    mp.spawn(overflow_worker, args=(2, free_port()), nprocs=2, join=True)


def test_shard_bookkeeping():
    from beta_recsys_amd.sharded import shard_rows

    for n in (1, 7, 8, 23, 6040):
        for world in (1, 2, 3, 8):
            sizes = [shard_rows(n, r, world) for r in range(world)]
            assert sum(sizes) == n
            assert sizes == [len(range(r, n, world)) for r in range(world)]


def test_sharded_engine_refuses_cpu_without_backend():
    """The product backend is HIP-only: constructing it on a CPU device fails loudly."""
    from beta_recsys_amd.sharded import HipKernels

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HipKernels(torch.device("cpu"))


def dp_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from beta_recsys_amd.replicated import allreduce_sum_

        U, I, D, b = 17, 13, 8, 12
        w = onp.init_params(U, I, D, seed=2)
        rng = np.random.default_rng(5)
        users, pos, neg = (rng.integers(0, n, world * b) for n in (U, I, I))
        sl = slice(rank * b, (rank + 1) * b)
        loss, reg, g = onp.mf_bpr_grads(w, users[sl], pos[sl], neg[sl], global_batch=world * b)
        buf = torch.from_numpy(np.concatenate([g[k].ravel() for k in KEYS] + [[loss, reg]]).astype(np.float32))
        allreduce_sum_(buf)
        if rank == 0:
            torch.save({"buf": buf.numpy(), "w": w, "batch": (users, pos, neg)}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_replicated_gradient_sum_equals_global_batch_gradient(tmp_path):
    """The identity the replicated engine rests on: sum over ranks of per-rank gradients scaled by
    the GLOBAL 1/B == the reference's gradient on the concatenated batch."""
    out_path = str(tmp_path / "dp.pt")
    mp.spawn(dp_worker, args=(2, free_port(), out_path), nprocs=2, join=True)
    res = torch.load(out_path, weights_only=False)
    loss, reg, g = onp.mf_bpr_grads(res["w"], *res["batch"])
    ref = np.concatenate([g[k].ravel() for k in KEYS] + [[loss, reg]])
    assert_tensor_close(res["buf"], ref, 1e-5, "all-reduced [grad | loss | reg]")


# ---- row-sharded NCF (sharded_ncf.py): host logic on 2 gloo ranks, the numpy oracle as the kernel backend ------

class OracleNcfKernels:
    """oracle/ncf_numpy.py behind the sharded NCF engine's kernel seam (CPU tensors)."""

    def __init__(self, kind):
        self.kind, self.st = kind, None

    def reset_clock(self, beta1, beta2):
        pass

    def gather_rows(self, table, idx):
        return table[idx].clone()

    def scatter_add_rows(self, table, idx, src):
        table.index_add_(0, idx, src.contiguous())

    def grad_on_fetched(self, model, g_flat, fetched, g_fetched, ratings, inv_batch, loss_out):
        from oracle import ncf_numpy as onc

        b = ratings.numel()
        names = dict(zip(("user_mlp", "item_mlp", "user_mf", "item_mf"), model._names()))
        w = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
        for field, t in fetched.items():          # the fetched rows play the tables: sample k reads row k
            w[names[field]] = t.numpy()
        loss, g, _ = onc.ncf_grads(w, np.arange(b), np.arange(b), ratings.numpy(), self.kind)
        scale = np.float32(b * inv_batch)         # the oracle averages over the LOCAL batch
        gv = model.views(g_flat)
        for k, v in g.items():
            field = next((f for f, n in names.items() if n == k), None)
            if field is not None:
                g_fetched[field] += torch.from_numpy(v * scale)
            else:
                gv[k] += torch.from_numpy(v * scale).reshape(gv[k].shape)
        loss_out[0] = float(loss) * float(scale)
        loss_out[1] = 0.0

    def advance_clock(self):
        pass

    def opt_step(self, opt, flat_w, flat_g):
        if self.st is None:
            self.st = onp.new_opt_state({"flat": flat_w.numpy()}, opt.name)
        onp.opt_step({"flat": flat_w.numpy()}, {"flat": flat_g.numpy()}, self.st, opt.name, opt.lr)
        flat_g.zero_()

    def check_status(self):
        pass


def ncf_worker(rank, world, port, kind, optimizer, lr, splits, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import beta_recsys_amd as hp
        from beta_recsys_amd import sharded_ncf

        U, I, E, L = 23, 19, 4, 2
        cfg = {"model": dict(n_users=U, n_items=I, emb_dim=E, dropout=0.0, device_str="cpu", optimizer=optimizer, lr=lr,
                             batch_size=8, model="ncf_end", mlp_config={"n_layers": L}, gmf_config={}),
               "system": {"run_dir": "/tmp/hiprec_test_runs"}}
        torch.manual_seed(3)   # the same seed on every rank: the same full model, every rank keeps its rows
        with contextlib.redirect_stdout(io.StringIO()):
            full = getattr(hp, {"neumf": "NeuMF", "gmf": "GMF", "mlp": "MLP"}[kind])(cfg["model"]).state_dict()
        engine_cls = {"neumf": sharded_ncf.ShardedNeuMFEngine, "gmf": sharded_ncf.ShardedGMFEngine,
                      "mlp": sharded_ncf.ShardedMLPEngine}[kind]
        eng = engine_cls(cfg, kernels=OracleNcfKernels(kind), full_state=full)
        rng = np.random.default_rng(100)
        losses, batches = [], []
        for split in splits:
            B = sum(split)
            users, items = rng.integers(0, U, B), rng.integers(0, I, B)
            items[: B // 3] = items[0]   # a popular item: many rows fetched from one owner
            ratings = (rng.random(B) < 0.3).astype(np.float32)
            batches.append((users, items, ratings))
            lo = sum(split[:rank])
            sl = slice(lo, lo + split[rank])
            losses.append(eng.train_single_batch(users[sl], items[sl], ratings[sl]))
        out = eng.gather_full_state_dict()
        if rank == 0:
            torch.save({"losses": losses, "full": {k: v.numpy() for k, v in out.items()}, "batches": batches,
                        "w0": {k: v.numpy() for k, v in full.items()}}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,optimizer,lr", [("neumf", "adam", 0.01), ("neumf", "sgd", 0.1), ("gmf", "adam", 0.01),
                                               ("mlp", "rmsprop", 0.01)])
def test_two_rank_sharded_ncf_equals_single_process(tmp_path, kind, optimizer, lr):
    """Tables row-sharded over 2 ranks (owner = row mod 2), tower replicated: the same losses and the same full
    state_dict as oracle/ncf_numpy.py's single-process step on the concatenated batches -- even / uneven splits,
    an empty rank, a popular item."""
    from oracle import ncf_numpy as onc

    splits = [(10, 10), (13, 7), (20, 0), (2, 3)]
    out_path = str(tmp_path / "out.pt")
    mp.spawn(ncf_worker, args=(2, free_port(), kind, optimizer, lr, splits, out_path), nprocs=2, join=True)
    res = torch.load(out_path, weights_only=False)
    w = {k: v.copy() for k, v in res["w0"].items()}
    st = onc.new_opt_state(w, optimizer)
    for batch, loss in zip(res["batches"], res["losses"]):
        ref = onc.ncf_train_step(w, st, batch, kind, optimizer, lr)
        assert_scalar_close(loss, ref, 2e-5, "loss")
    from helpers import assert_ncf_end_state

    assert set(res["full"]) == set(w) and all(res["full"][k].shape == w[k].shape for k in w)
    assert_ncf_end_state(res["full"], res["w0"], res["batches"], kind, optimizer, lr, f"sharded {kind}")


# ---- epoch-planned sharded SGD (ShardedMFEngine.plan_epoch / run_planned_epoch) on gloo ------------------------------

class OraclePlannedKernels(OracleKernels):
    """numpy statements of the epoch planner (csrc/plan.hip: tests/plan_statements.py) and of the planned step's
    kernels (csrc/mf_owned.hip REMOTE variants, csrc/shard.hip)."""

    status = 0

    def plan_route(self, users, pos, neg, perm, bs, world, n_users, n_items):
        send, cnt_ds, status = ps.plan_route(users.numpy(), pos.numpy(), neg.numpy(),
                                             None if perm is None else perm.numpy(), bs, world, n_users, n_items)
        self.status |= status
        return torch.from_numpy(send), torch.from_numpy(cnt_ds)

    def plan_status(self):
        return torch.tensor([self.status], dtype=torch.int64)

    def check_status(self):
        if self.status:
            self.status = 0
            raise IndexError("index out of range in self")
        super().check_status()

    def clear_status(self):
        self.status = 0

    def plan_place_triples(self, recv, recv_cnt, S, cap):
        return tuple(torch.from_numpy(a) for a in ps.plan_place_triples(recv.numpy(), recv_cnt.numpy(), S, cap)) + (None,)

    def plan_item_slots(self, U, P, N, S, cap, world, n_users_local, n_items, fill=None):
        out = ps.plan_item_slots(U.numpy(), P.numpy(), N.numpy(), S, cap, world, n_users_local)
        ps.check_item_slots(out, U.numpy(), P.numpy(), N.numpy(), S, cap, world, n_users_local)
        return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in out.items()}

    def plan_place_requests(self, incoming, in_qs, S, n_rows_local=0):
        return tuple(torch.from_numpy(a) for a in ps.plan_place_requests(incoming.numpy(), in_qs.numpy(), S))

    def payload_zero(self, item_emb, item_bias, idx, payload, g_send):
        idx = idx.long()
        payload.copy_(torch.cat([self.gather_rows(item_emb, idx), self.gather_rows(item_bias, idx)], dim=1))
        g_send.zero_()

    def _remote_grads(self, model, fetched, users, slot_pos, slot_neg, own, total, inv_batch, reg_coef):
        D = model.emb_dim
        live = users >= 0
        # the ownership arrays must describe the batch: -1 / total 1 exactly for rows referenced once
        for role, ids in ((0, users), (1, slot_pos), (2, slot_neg)):
            pool = users[live] if role == 0 else torch.cat([slot_pos[live], slot_neg[live]])
            for t in torch.nonzero(live).flatten().tolist():
                cnt = int((pool == ids[t]).sum())
                s = int(own[role][t])
                assert (cnt == 1) == (s < 0 or int(total[s]) == 1), "ownership does not match the batch"
                assert s < 0 or int(total[s]) == cnt
        if not bool(live.any()):   # this rank owns none of the step's users
            self.partial = torch.zeros(3)
            return None
        ue, ie, ub, ib, gb = model._views(model.flat)
        w = {"user_emb.weight": ue.numpy(), "user_bias.weight": ub.numpy(), "global_bias": gb.numpy(),
             "item_emb.weight": fetched[:, :D].numpy(), "item_bias.weight": fetched[:, D:].numpy()}
        loss, reg, g = onp.mf_bpr_grads(w, users[live].numpy(), slot_pos[live].numpy(), slot_neg[live].numpy(),
                                        reg_coef, global_batch=int(round(1.0 / inv_batch)))
        self.partial = torch.tensor([loss, reg, float(g["global_bias"][0])], dtype=torch.float32)
        return g

    def owned_remote_step(self, model, fetched, g_send, n_slots, users, slot_pos, slot_neg, own, total, arrived, acc,
                          inv_batch, reg_coef, lr):
        g = self._remote_grads(model, fetched, users, slot_pos, slot_neg, own, total, inv_batch, reg_coef)
        if g is None:
            return
        D = model.emb_dim
        ue, ie, ub, ib, gb = model._views(model.flat)
        lr32 = np.float32(lr)
        ue -= torch.from_numpy(lr32 * g["user_emb.weight"])      # untouched rows have zero gradient
        ub -= torch.from_numpy(lr32 * g["user_bias.weight"])
        g_send[:, :D] += torch.from_numpy(g["item_emb.weight"])
        g_send[:, D:] += torch.from_numpy(g["item_bias.weight"])

    def grad_remote_step(self, model, g_flat, fetched, g_send, n_slots, users, slot_pos, slot_neg, own, total,
                         inv_batch, reg_coef):
        g = self._remote_grads(model, fetched, users, slot_pos, slot_neg, own, total, inv_batch, reg_coef)
        if g is None:
            return
        D = model.emb_dim
        gue, gie, gub, gib, ggb = model._views(g_flat)
        gue += torch.from_numpy(g["user_emb.weight"])
        gub += torch.from_numpy(g["user_bias.weight"])
        g_send[:, :D] += torch.from_numpy(g["item_emb.weight"])
        g_send[:, D:] += torch.from_numpy(g["item_bias.weight"])

    def publish_partials(self, g_send, dim, extra_rows):
        g_send[extra_rows.long(), :3] = self.partial

    def apply_finish(self, t_emb, t_bias, idx, g_recv, coef, extra_pos, scalar_target, scalar_coef, first_of_epoch):
        idx = idx.long()
        keep = idx >= 0
        t_emb.index_add_(0, idx[keep], np.float32(coef) * g_recv[keep][:, :-1])
        t_bias.index_add_(0, idx[keep], np.float32(coef) * g_recv[keep][:, -1:])
        tot = g_recv[extra_pos.long(), :3].sum(0)
        if first_of_epoch:
            self.sums = [0.0, 0.0]
        self.loss, self.reg = float(tot[0]), float(tot[1])
        self.sums[0] += self.loss
        self.sums[1] += self.reg
        scalar_target += np.float32(scalar_coef) * tot[2]

    def epoch_stats(self):
        return self.loss, self.reg, self.sums[0], self.sums[1]


def planned_worker(rank, world, port, n_local, bs, shuffle, out_path, optimizer="sgd", lr=0.1, U=37, I=23):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import beta_recsys_amd as hp
        from beta_recsys_amd.sharded import ShardedMFEngine

        D = 8    # 37 % 4 != 0, 23 % 4 != 0 (and % 8): uneven shards
        w0 = onp.init_params(U, I, D, seed=7)
        rng = np.random.default_rng(50 + rank)
        users, neg = rng.integers(0, U, n_local), rng.integers(0, I, n_local)
        p = 1.0 / np.arange(1, I + 1)
        pos = rng.choice(I, n_local, p=p / p.sum())            # Zipf items: duplicates inside a step's requests
        with contextlib.redirect_stdout(io.StringIO()):
            eng = ShardedMFEngine(make_config(U, I, D, optimizer, lr, "padded", "rows"), kernels=OraclePlannedKernels(),
                                  full_state={k: torch.from_numpy(v) for k, v in w0.items()})
        loader = hp.DeviceTripleBatcher(torch.from_numpy(users), torch.from_numpy(pos), torch.from_numpy(neg), bs,
                                        shuffle=shuffle, generator=torch.Generator().manual_seed(9 + rank) if shuffle else None)
        # world 3: the plan's exchanges run over a process group of their own, as prefetch_plan does on GPUs
        plan = eng.plan_epoch(loader, group=dist.new_group() if world == 3 else None)
        # what the plan says this rank's local batches were (the order inside a batch is irrelevant)
        order = np.arange(n_local)
        if shuffle:
            order = torch.randperm(n_local, generator=torch.Generator().manual_seed(9 + rank)).numpy()
        # requests are de-duplicated: never more slots than distinct (item, owner) pairs, at most 2 per triple
        for s in range(plan["S"]):
            blk = slice(s * plan["cap"], (s + 1) * plan["cap"])
            live = plan["U"][blk] >= 0
            refs = torch.cat([plan["SP"][blk][live], plan["SN"][blk][live]])
            assert plan["n_slots"][s] - world == refs.unique().numel() <= 2 * int(live.sum())
        stats = eng.run_planned_epoch(plan)
        full = eng.gather_full_state_dict()
        gathered = [None] * world
        dist.all_gather_object(gathered, (users[order], pos[order], neg[order]))
        if rank == 0:
            torch.save({"stats": stats, "full": {k: v.numpy() for k, v in full.items()}, "w0": w0,
                        "local": gathered, "bs": bs}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def check_planned(res, n_local, bs, optimizer, lr):
    from helpers import assert_on_trajectory, assert_sgd_exact, mf_trajectory

    batches = [tuple(np.concatenate([loc[j][k:k + bs] for loc in res["local"]]) for j in range(3))
               for k in range(0, n_local, bs)]
    w = onp.copy_params(res["w0"])
    st = onp.new_opt_state(w, optimizer)
    tot_loss = tot_reg = 0.0
    for batch in batches:
        loss, reg = onp.mf_train_step(w, st, batch, "bpr", optimizer, lr)
        tot_loss += loss
        tot_reg += reg
    last_loss, last_reg, loss_sum, reg_sum = res["stats"]
    assert_scalar_close(last_loss, loss, 2e-5, "last step's global loss")
    assert_scalar_close(loss_sum, tot_loss, 2e-5, "epoch loss sum")
    assert_scalar_close(reg_sum, tot_reg, 2e-5, "epoch regularizer sum")
    for k in KEYS:
        assert res["full"][k].shape == w[k].shape
    if optimizer == "sgd":
        assert_sgd_exact(res["full"], w, res["w0"], "after the planned epoch", lr=lr, batch=bs * len(res["local"]))
    else:
        w_ref, env, upd = mf_trajectory(res["w0"], batches, optimizer, lr)
        assert_on_trajectory(res["full"], w_ref, env, upd, "after the planned epoch")


@pytest.mark.parametrize("world,n_local,bs,shuffle", [(2, 50, 16, False), (4, 41, 8, True), (3, 20, 32, False)])
def test_planned_sharded_epoch_equals_single_process(tmp_path, world, n_local, bs, shuffle):
    """plan_epoch + run_planned_epoch on 2, 3 and 4 gloo ranks with uneven shards (n_rows % R != 0), Zipf items and a
    short last batch: the epoch-level routing (triples to owner(user), de-duplicated item requests to owner(item)),
    the per-step exact-size exchanges and the partials riding in the extra rows reproduce the single-process SGD
    steps on the concatenated global batches -- loss sums and the gathered full state_dict, every element within
    1e-5 of the update."""
    out_path = str(tmp_path / "out.pt")
    mp.spawn(planned_worker, args=(world, free_port(), n_local, bs, shuffle, out_path), nprocs=world, join=True)
    check_planned(torch.load(out_path, weights_only=False), n_local, bs, "sgd", 0.1)


@pytest.mark.parametrize("world,optimizer,lr", [(2, "adam", 0.05), (4, "adam", 0.05), (2, "rmsprop", 0.01)])
def test_planned_sharded_epoch_with_the_dense_optimizers(tmp_path, world, optimizer, lr):
    """Adam / RMSprop on the planned path (torch_engine.py:30-39): the same routing and exchanges, the
    gradient kernel's dense form (user rows -> the shard's dense gradient, item slots -> the exchange), the owners
    accumulate what comes back into their dense gradient, one local sweep per step.  Against the single-process
    oracle on the concatenated batches, every element inside the legal-trajectory envelope."""
    n_local, bs = 41, 8
    out_path = str(tmp_path / "out.pt")
    mp.spawn(planned_worker, args=(world, free_port(), n_local, bs, True, out_path, optimizer, lr), nprocs=world,
             join=True)
    check_planned(torch.load(out_path, weights_only=False), n_local, bs, optimizer, lr)


def test_planned_sharded_epoch_on_eight_ranks(tmp_path):
    """World size 8 (the node BASELINE configs[3] names) with Zipf items, 37 users and 23 items (n_rows % 8 != 0:
    shards of 5 and 4 / 3 and 2 rows), a short last batch and steps in which some ranks receive nothing."""
    n_local, bs = 21, 8
    out_path = str(tmp_path / "out.pt")
    mp.spawn(planned_worker, args=(8, free_port(), n_local, bs, True, out_path), nprocs=8, join=True)
    check_planned(torch.load(out_path, weights_only=False), n_local, bs, "sgd", 0.1)


# ---- data-parallel replicas of the graph / sibling engines: the identity they rest on, on 2 gloo ranks ---------------

def flat_dp_worker(rank, world, port, model, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from beta_recsys_amd.replicated import DirectAllReduce

        share = np.float32(1.0 / world)     # FlatModelEngine._batch_share() of a replica
        rng = np.random.default_rng(3)
        U, I, D, B = 40, 30, 8, 24
        users, pos, neg = rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)
        sl = slice(rank * B // world, (rank + 1) * B // world)
        if model == "lightgcn":
            from oracle import lightgcn_numpy as olg

            adj = olg.build_norm_adj(U, I, rng.integers(0, U, 200), rng.integers(0, I, 200))
            w = {"user_embedding.weight": rng.standard_normal((U, D)).astype(np.float32) * 0.3,
                 "item_embedding.weight": rng.standard_normal((I, D)).astype(np.float32) * 0.3}
            full = olg.lightgcn_grads(w, adj, 2, users, pos, neg, 1e-3)
            loss, g = olg.lightgcn_grads(w, adj, 2, users[sl], pos[sl], neg[sl], 1e-3)   # mean over the LOCAL batch
            part = [loss * share] + [g[k] * share for k in olg.KEYS]
            ref = [full[0]] + [full[1][k] for k in olg.KEYS]
        elif model == "ngcf":
            from oracle import ngcf_numpy as ong
            from oracle import lightgcn_numpy as olg

            adj = olg.build_norm_adj(U, I, rng.integers(0, U, 200), rng.integers(0, I, 200))
            w = {"user_embedding.weight": rng.standard_normal((U, D)).astype(np.float32) * 0.3,
                 "item_embedding.weight": rng.standard_normal((I, D)).astype(np.float32) * 0.3}
            for l in range(2):
                for kind in ("GC", "Bi"):
                    w[f"{kind}_weights.{l}.weight"] = rng.standard_normal((D, D)).astype(np.float32) * 0.3
                    w[f"{kind}_weights.{l}.bias"] = rng.standard_normal(D).astype(np.float32) * 0.1
            w = {k: w[k] for k in ong.keys(2)}
            masks = [rng.random((U + I, D)) < 0.9 for _ in range(2)]     # the SAME message dropout on every replica
            full = ong.ngcf_grads(w, adj, users, pos, neg, 1e-3, B, masks, [0.1, 0.1])
            # the regulariser divides by the CONFIGURED (global) batch size on every replica (ngcf.py:189); only the
            # BPR mean is over the local batch and takes the 1 / world share
            l_loc, g_loc = ong.ngcf_grads(w, adj, users[sl], pos[sl], neg[sl], 0.0, B, masks, [0.1, 0.1])
            l_reg, g_reg = ong.ngcf_grads(w, adj, users[sl], pos[sl], neg[sl], 1e-3, B, masks, [0.1, 0.1])
            part = [l_loc * share + (l_reg - l_loc)] + [g_loc[k] * share + (g_reg[k] - g_loc[k]) for k in ong.keys(2)]
            ref = [full[0]] + [full[1][k] for k in ong.keys(2)]
        else:   # pgmf: lambda ||v|| once (rank 0), clip AFTER the reduction
            from oracle import pgmf_numpy as opg

            w = {"user_memory.weight": rng.standard_normal((U, D)).astype(np.float32),
                 "item_memory.weight": rng.standard_normal((I, D)).astype(np.float32),
                 "v.weight": rng.standard_normal((1, D)).astype(np.float32)}
            keys = list(w)
            l_full, g_full = opg.pgmf_grads(w, users, pos, neg, 1e-2)
            opg.clip_grad_norm(g_full, 0.05)
            loss, g = opg.pgmf_grads(w, users[sl], pos[sl], neg[sl], 1e-2 if rank == 0 else 0.0)
            # the lambda term is not a batch mean: take it out before scaling, add it back on rank 0
            l0, g0 = opg.pgmf_grads(w, users[sl], pos[sl], neg[sl], 0.0)
            part = [l0 * share + (loss - l0)] + [g0[k] * share + (g[k] - g0[k]) for k in keys]
            ref = [l_full] + [g_full[k] for k in keys]
        buf = torch.from_numpy(np.concatenate([np.ravel(np.float32(x)) for x in part]))
        DirectAllReduce(None, torch.device("cpu")).sum_(buf)
        got = buf.numpy()
        if model == "pgmf":     # clip the reduced gradient, as the replicas do
            gdict, off = {}, 1
            for k in keys:
                gdict[k] = got[off:off + w[k].size].reshape(w[k].shape).copy()
                off += w[k].size
            opg.clip_grad_norm(gdict, 0.05)
            got = np.concatenate([got[:1]] + [gdict[k].ravel() for k in keys])
        if rank == 0:
            torch.save({"got": got, "ref": np.concatenate([np.ravel(np.float32(x)) for x in ref])}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model", ["lightgcn", "ngcf", "pgmf"])
def test_flat_replicas_sum_to_the_global_batch_gradient(tmp_path, model):
    """What replicated_flat_engine rests on, with the oracles as the kernels: per-rank gradients of the rank's share,
    batch means scaled by 1 / (local batch x world), the terms that are NOT batch means handled as the engines
    handle them (NGCF's regulariser over the configured batch on every replica, PairwiseGMF's lambda ||v|| on rank 0,
    its clip after the reduction), summed by the product's collective helper == the reference's single-process
    [loss | gradient] on the whole batch."""
    out_path = str(tmp_path / "dp.pt")
    mp.spawn(flat_dp_worker, args=(2, free_port(), model, out_path), nprocs=2, join=True)
    res = torch.load(out_path, weights_only=False)
    assert_tensor_close(res["got"][:1], res["ref"][:1], 2e-5, "loss")
    assert_tensor_close(res["got"][1:], res["ref"][1:], 2e-5, "all-reduced gradient")
