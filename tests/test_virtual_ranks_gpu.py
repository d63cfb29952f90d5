"""The N > 1 code of the multi-GPU engines, EXECUTED on one MI355X: R virtual ranks (host threads, each with its own
stream, engine and shard) over the loopback stand-ins of tests/loopback.py / tests/native/loopback_rccl.hip.

What runs here that no world-size-1 test reaches: the grouped ncclSend / ncclRecv branches of
``hiprec_shard_planned_steps`` (csrc/shard.hip: row exchange, gradient exchange, the partials riding in the extra
rows, offsets of the peers' segments), the planner's exchanges between ranks with the REAL planner kernels, the
``ncclAllReduce`` call inside ``hiprec_mf_bpr_dp_epoch_fused_range``, ``DirectAllReduce``, and the per-step
all-to-all routing of the sharded MF / NCF engines -- against the single-process oracle on the concatenated global
batches (SURVEY.md §8e "Semantics vs single-process reference")."""
import contextlib
import io

import numpy as np
import pytest
import torch

from helpers import REL  # noqa: F401  (north_star: 1e-5 relative)
from helpers import (KEYS, assert_mf_end_state, assert_ncf_end_state, assert_on_trajectory, assert_scalar_close,
                     assert_sgd_exact, mf_trajectory)
from loopback import VirtualWorld
from oracle import mf_numpy as onp

pytestmark = pytest.mark.gpu
RUN_DIR = {"run_dir": "/tmp/hiprec_test_runs", "model_save_dir": "/tmp/hiprec_test_runs"}


def zipf(rng, n_items, n):
    p = 1.0 / np.arange(1, n_items + 1)
    return rng.choice(n_items, n, p=p / p.sum())


def local_triples(rank, world, U, I, n_local, idle_rank=None):
    """One rank's share of the interaction stream: uniform users, Zipf positives (duplicates inside a step's
    requests), uniform negatives.  idle_rank: no user row of that rank is ever drawn -- it receives no triple in any
    step, yet owns item rows its peers ask for."""
    rng = np.random.default_rng(50 + rank)
    users = rng.integers(0, U, n_local)
    if idle_rank is not None:
        users = np.where(users % world == idle_rank, (users + 1) % U, users)
        users = np.where(users % world == idle_rank, (users + 1) % U, users)   # U % world != 0: the wrap can land there
    return users, zipf(rng, I, n_local), rng.integers(0, I, n_local)


def planned_rank(group, w0, U, I, D, n_local, bs, shuffle, optimizer, lr, driver, idle_rank, epochs=1, prefetch=False,
                 dense_opt="sweep", shard_sgd="pull", self_exchange=False):
    import beta_recsys_amd as hp
    from beta_recsys_amd.sharded import ShardedMFEngine

    rank, world = group.rank(), group.size()
    users, pos, neg = local_triples(rank, world, U, I, n_local, idle_rank)
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer=optimizer, lr=lr, batch_size=bs,
                         loss="bpr", sgd_mode="rows", step_driver=driver, dense_opt=dense_opt, shard_sgd=shard_sgd,
                         shard_self_exchange=self_exchange),
           "system": RUN_DIR}
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ShardedMFEngine(cfg, process_group=group, full_state={k: torch.from_numpy(v) for k, v in w0.items()})
    assert (eng._lazy is not None) == (dense_opt == "lazy" and optimizer != "sgd")
    gen = torch.Generator().manual_seed(9 + rank) if shuffle else None
    loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a).cuda() for a in (users, pos, neg)), bs, shuffle=shuffle,
                                    generator=gen)
    replay = torch.Generator().manual_seed(9 + rank)
    visited, stats = [], None
    for epoch in range(epochs):
        order = torch.randperm(n_local, generator=replay).numpy() if shuffle else np.arange(n_local)
        visited.append((users[order], pos[order], neg[order]))
        if prefetch:
            plan = eng.take_plan(loader)          # epoch 0: a fresh plan; later: the one prefetched during the last epoch
            if epoch + 1 < epochs:
                eng.prefetch_plan(loader)
        else:
            plan = eng.plan_epoch(loader)
        received = int((plan["U"] >= 0).sum())
        # plain SGD through the C driver runs as owner pulls (contribution lists in the plan) unless told otherwise
        assert (plan.get("contrib") is not None) == (optimizer == "sgd" and driver == "c" and shard_sgd == "pull"
                                                     and D % 4 == 0)
        stats = eng.run_planned_epoch(plan)
    full = eng.gather_full_state_dict()
    pb = eng._planned_bufs
    clean = optimizer != "sgd" or (float(pb["acc"].abs().max()) == 0.0 and int(pb["arrived"].abs().max()) == 0)
    return {"stats": stats, "full": {k: v.cpu().numpy() for k, v in full.items()} if rank == 0 else None,
            "visited": visited, "mode": eng._step_mode, "clean": clean, "received": received,
            "g_zero": optimizer == "sgd" or float(eng._g_flat.abs().max()) == 0.0}


def check_planned(res, w0, n_local, bs, optimizer, lr, epochs=1):
    world = len(res)
    batches = []
    for e in range(epochs):
        batches += [tuple(np.concatenate([r["visited"][e][j][k:k + bs] for r in res]) for j in range(3))
                    for k in range(0, n_local, bs)]
    per_epoch = len(batches) // epochs
    w = onp.copy_params(w0)
    st = onp.new_opt_state(w, optimizer)
    tot_loss = tot_reg = 0.0
    for i, batch in enumerate(batches):
        if i % per_epoch == 0:
            tot_loss = tot_reg = 0.0
        loss, reg = onp.mf_train_step(w, st, batch, "bpr", optimizer, lr)
        tot_loss += loss
        tot_reg += reg
    for r in res:     # every rank reads the GLOBAL loss: the partials rode in the extra rows of the gradient exchange
        last_loss, last_reg, loss_sum, reg_sum = r["stats"]
        assert_scalar_close(last_loss, loss, REL, "last step's global loss")
        assert_scalar_close(loss_sum, tot_loss, REL, "epoch loss sum")
        assert_scalar_close(reg_sum, tot_reg, REL, "epoch regularizer sum")
        assert r["clean"], "the owned-rows accumulators are left zeroed"
        assert r["g_zero"], "the sweep leaves the dense gradient zeroed"
    full = res[0]["full"]
    for k in KEYS:
        assert full[k].shape == w[k].shape
    if optimizer == "sgd":
        assert_sgd_exact(full, w, w0, f"planned epoch on {world} virtual ranks", lr=lr, batch=bs * world)
    else:
        w_ref, env, upd = mf_trajectory(w0, batches, optimizer, lr)
        assert_on_trajectory(full, w_ref, env, upd, f"planned epoch on {world} virtual ranks")


@pytest.mark.parametrize("world,D,bs,optimizer,lr,shuffle,idle", [
    (2, 64, 256, "sgd", 0.05, False, None), (2, 64, 256, "adam", 0.05, True, None),
    (2, 128, 300, "rmsprop", 0.01, False, None), (4, 128, 300, "sgd", 0.05, True, 1),
    (4, 64, 256, "adam", 0.05, False, 2), (4, 100, 200, "rmsprop", 0.01, True, None),
    (8, 64, 128, "sgd", 0.05, True, 5), (8, 128, 128, "adam", 0.05, True, None),
    (8, 2, 64, "rmsprop", 0.01, False, 0), (3, 64, 200, "sgd", 0.05, True, None)])
def test_planned_steps_exchange_between_virtual_ranks(hip_device, world, D, bs, optimizer, lr, shuffle, idle):
    """``hiprec_shard_planned_steps`` with R > 1: ONE C call per epoch posts, per step, the grouped send / recv of the
    requested rows, launches the gradient kernel on (local users, fetched slots), and posts the grouped send / recv of
    the slot gradients with every peer's loss partials in the extra rows.  n_rows % R != 0 (3001 users, 403 items),
    Zipf items, a short last batch, a rank that receives no triple at all; SGD every element within 1e-5 of the
    update, Adam / RMSprop inside the legal-trajectory envelope of the single-process oracle."""
    U, I = 3001, 403
    n_local = 4 * bs + bs // 3
    w0 = onp.init_params(U, I, D, seed=3)
    vw = VirtualWorld(world)
    try:
        res = vw.run(lambda g: planned_rank(g, w0, U, I, D, n_local, bs, shuffle, optimizer, lr, "c", idle))
        counters = vw.counters()
    finally:
        vw.close()
    assert all(r["mode"] == "c" for r in res)
    assert sum(r["received"] for r in res) == world * n_local
    if idle is not None:
        assert res[idle]["received"] == 0 and all(r["received"] > 0 for i, r in enumerate(res) if i != idle)
    steps = (n_local + bs - 1) // bs
    # two grouped exchanges per step and rank, each with one send and one recv per peer
    assert counters["groups"] == 2 * steps * world, counters
    assert counters["sends"] == counters["recvs"] == 2 * steps * world * (world - 1), counters
    assert counters["bytes"] > 0
    check_planned(res, w0, n_local, bs, optimizer, lr)


@pytest.mark.parametrize("world,optimizer,lr,dense_opt", [(2, "sgd", 0.05, "sweep"), (3, "adam", 0.05, "lazy"),
                                                          (4, "rmsprop", 0.01, "sweep")])
def test_planned_steps_with_the_own_segment_exchanged_too(hip_device, world, optimizer, lr, dense_opt):
    """HIPREC_SHARD_EXCHANGE_SELF (`shard_self_exchange`) on a world of several ranks: every rank's own segment of both
    exchanges goes through the communicator as well -- one more send and one more recv per rank, exchange and step,
    posted to itself inside the same group -- and the epoch is the oracle's like the in-place form's."""
    U, I, D, bs = 3001, 403, 64, 256
    n_local = 3 * bs + bs // 3
    w0 = onp.init_params(U, I, D, seed=3)
    vw = VirtualWorld(world)
    try:
        res = vw.run(lambda g: planned_rank(g, w0, U, I, D, n_local, bs, True, optimizer, lr, "c", None,
                                            dense_opt=dense_opt, self_exchange=True))
        counters = vw.counters()
    finally:
        vw.close()
    steps = (n_local + bs - 1) // bs
    assert all(r["mode"] == "c" for r in res)
    assert counters["groups"] == 2 * steps * world and counters["sends"] == counters["recvs"] == 2 * steps * world * world
    check_planned(res, w0, n_local, bs, optimizer, lr)


@pytest.mark.parametrize("world,D,bs", [(2, 64, 256), (4, 128, 300)])
def test_planned_sgd_steps_in_the_atomic_form_between_virtual_ranks(hip_device, world, D, bs):
    """`shard_sgd: "atomic"`: the round 2-4 step (shared user rows and shared item slots through device-scope atomics,
    the partials' publish as a launch of its own) still is what it was -- the default since round 5 is the owner-pulls
    step, which the tests above run."""
    U, I = 3001, 403
    n_local = 4 * bs + bs // 3
    w0 = onp.init_params(U, I, D, seed=3)
    vw = VirtualWorld(world)
    try:
        res = vw.run(lambda g: planned_rank(g, w0, U, I, D, n_local, bs, True, "sgd", 0.05, "c", None,
                                            shard_sgd="atomic"))
    finally:
        vw.close()
    assert all(r["mode"] == "c" for r in res)
    check_planned(res, w0, n_local, bs, "sgd", 0.05)


@pytest.mark.parametrize("world,optimizer,lr,driver", [(4, "adam", 0.05, "c"), (8, "rmsprop", 0.01, "c"),
                                                       (2, "adam", 0.05, "torch")])
def test_lazy_optimizer_state_between_virtual_ranks(hip_device, world, optimizer, lr, driver):
    """Exact lazy Adam / RMSprop (csrc/lazy_opt.hip) on the row-sharded planned path with R > 1: every owner catches
    the item rows its peers ask for up before it serves them, the user rows before the gradient kernel reads them, and
    steps the step's rows when the exchanged gradients have arrived -- two epochs, one rank idle, against the oracle's
    dense optimizer on the global batches."""
    U, I, D, bs, epochs = 3001, 403, 64, 200, 2
    n_local = 3 * bs + 31
    w0 = onp.init_params(U, I, D, seed=3)
    vw = VirtualWorld(world)
    try:
        res = vw.run(lambda g: planned_rank(g, w0, U, I, D, n_local, bs, True, optimizer, lr, driver, 1, epochs=epochs,
                                            dense_opt="lazy"))
    finally:
        vw.close()
    check_planned(res, w0, n_local, bs, optimizer, lr, epochs=epochs)


@pytest.mark.parametrize("optimizer,lr", [("sgd", 0.05), ("adam", 0.05)])
def test_planned_steps_through_the_torch_loop_between_virtual_ranks(hip_device, optimizer, lr):
    """The same epoch through the torch.distributed-shaped loop (``step_driver: "torch"``): the per-step launches the
    C driver makes, called one by one around all_to_all_single -- the two drivers must agree with the same oracle."""
    world, U, I, D, bs = 4, 3001, 403, 64, 256
    n_local = 3 * bs + 17
    w0 = onp.init_params(U, I, D, seed=3)
    vw = VirtualWorld(world)
    try:
        res = vw.run(lambda g: planned_rank(g, w0, U, I, D, n_local, bs, True, optimizer, lr, "torch", None))
        counters = vw.counters()
    finally:
        vw.close()
    assert all(r["mode"] == "torch" for r in res) and counters["sends"] == 0
    check_planned(res, w0, n_local, bs, optimizer, lr)


def test_planned_epochs_with_prefetched_plans_between_virtual_ranks(hip_device):
    """Three epochs on 4 virtual ranks, the plan of epoch k + 1 made on the side stream over a group of its own while
    epoch k's steps (already enqueued from C) run: the hand-over between streams and the exchanges of plan and steps
    interleaving between ranks leave the oracle's state."""
    world, U, I, D, bs, epochs = 4, 3001, 403, 64, 256, 3
    n_local = 3 * bs + 40
    w0 = onp.init_params(U, I, D, seed=3)
    vw = VirtualWorld(world)
    try:
        res = vw.run(lambda g: planned_rank(g, w0, U, I, D, n_local, bs, True, "sgd", 0.05, "c", None, epochs=epochs,
                                            prefetch=True))
    finally:
        vw.close()
    check_planned(res, w0, n_local, bs, "sgd", 0.05, epochs=epochs)


def test_planned_steps_at_a_configs3_like_batch_between_virtual_ranks(hip_device):
    """4 virtual ranks, dim 128, 16 384 triples per rank and step over 400 k x 40 k ids: thousands of slots per
    exchange segment, shared slots, rows several peers return gradients for (the atomic path of the apply kernel)
    and rows only one does (the plain read-modify-write path)."""
    world, U, I, D, bs = 4, 400_003, 40_001, 128, 16_384
    n_local = 2 * bs + 1000
    w0 = onp.init_params(U, I, D, seed=5)
    vw = VirtualWorld(world, timeout=60.0)
    try:
        res = vw.run(lambda g: planned_rank(g, w0, U, I, D, n_local, bs, False, "sgd", 0.05, "c", None))
        counters = vw.counters()
    finally:
        vw.close()
    assert counters["bytes"] > 3 * 2 * world * 1000 * (D + 1) * 4      # real volumes, not a handful of rows
    check_planned(res, w0, n_local, bs, "sgd", 0.05)


@pytest.mark.parametrize("consistent", [True, False])
def test_a_mis_sized_plan_fails_instead_of_hanging(hip_device, consistent):
    """Rank 0's plan claims one row less from rank 1 in step 1 than rank 1 will send.  If the plan still adds up on
    rank 0 (its block size changed with it) only the exchange can notice: the loopback pairs the recv with the send,
    sees the sizes differ, and the C driver returns an error naming the exchange on every rank -- within the
    rendezvous time-out, never a hang.  If it does not even add up locally the driver refuses the whole range BEFORE
    it posts anything, and the peers are released by the abort."""
    import beta_recsys_amd as hp
    from beta_recsys_amd.sharded import ShardedMFEngine

    world, U, I, D, bs = 2, 301, 97, 16, 64
    w0 = onp.init_params(U, I, D, seed=3)

    def rank_fn(group):
        rank = group.rank()
        users, pos, neg = local_triples(rank, world, U, I, 3 * bs)
        cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer="sgd", lr=0.05,
                             batch_size=bs, loss="bpr", sgd_mode="rows"), "system": RUN_DIR}
        with contextlib.redirect_stdout(io.StringIO()):
            eng = ShardedMFEngine(cfg, process_group=group, full_state={k: torch.from_numpy(v) for k, v in w0.items()})
        loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a).cuda() for a in (users, pos, neg)), bs, shuffle=False)
        plan = eng.plan_epoch(loader)
        if rank == 0:
            assert plan["req_cnt_h"][1][1] > 0
            plan["req_cnt_h"][1][1] -= 1
            if consistent:
                plan["n_slots_h"][1] -= 1
        eng.run_planned_epoch(plan)

    vw = VirtualWorld(world, timeout=5.0)
    try:
        with pytest.raises(RuntimeError, match="row exchange of step 1" if consistent else "inconsistent plan"):
            vw.run(rank_fn)
        assert vw.failed()
        if consistent:
            assert "which sends" in vw.last_error()
    finally:
        vw.close()


def test_an_out_of_range_id_on_one_rank_raises_on_every_rank(hip_device):
    """The planner kernels flag rank 2's bad item id in ITS status word; the word is OR-ed over the group, so all four
    ranks raise IndexError between the same two collectives, and all four plan and run the corrected epoch."""
    import beta_recsys_amd as hp
    from beta_recsys_amd.sharded import ShardedMFEngine

    world, U, I, D, bs = 4, 301, 97, 16, 64
    w0 = onp.init_params(U, I, D, seed=3)

    def rank_fn(group):
        rank = group.rank()
        users, pos, neg = local_triples(rank, world, U, I, 3 * bs)
        cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer="sgd", lr=0.05,
                             batch_size=bs, loss="bpr", sgd_mode="rows"), "system": RUN_DIR}
        with contextlib.redirect_stdout(io.StringIO()):
            eng = ShardedMFEngine(cfg, process_group=group, full_state={k: torch.from_numpy(v) for k, v in w0.items()})
        bad = pos.copy()
        if rank == 2:
            bad[7] = I
        loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a).cuda() for a in (users, bad, neg)), bs, shuffle=False)
        with pytest.raises(IndexError, match="item index"):
            eng.plan_epoch(loader)
        loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a).cuda() for a in (users, pos, neg)), bs, shuffle=False)
        return eng.run_planned_epoch(eng.plan_epoch(loader))

    vw = VirtualWorld(world, timeout=20.0)
    try:
        res = vw.run(rank_fn)
    finally:
        vw.close()
    assert all(np.isfinite(r[2]) for r in res)


def dp_rank(group, w0, U, I, D, n_local, bs, optimizer, lr, epochs, collective="rccl"):
    import beta_recsys_amd as hp
    from beta_recsys_amd.replicated import ReplicatedMFEngine

    rank, world = group.rank(), group.size()
    users, pos, neg = local_triples(rank, world, U, I, n_local)
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer=optimizer, lr=lr, batch_size=bs,
                         loss="bpr", dp_collective=collective, prefetch_epoch=False), "system": RUN_DIR}
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ReplicatedMFEngine(cfg, process_group=group)
    eng.model.load_state_dict({k: torch.from_numpy(v) for k, v in w0.items()})
    loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a).cuda() for a in (users, pos, neg)), bs, shuffle=False)
    sums = []
    with contextlib.redirect_stdout(io.StringIO()):
        for epoch in range(epochs):
            eng.train_an_epoch(loader, epoch)
            sums.append((eng.writer.scalars[-2][1], eng.writer.scalars[-1][1]))
    return {"sums": sums, "w": {k: v.cpu().numpy() for k, v in eng.model.state_dict().items()},
            "local": (users, pos, neg), "direct": eng._direct_comm is not None}


@pytest.mark.parametrize("world,optimizer,lr", [(2, "adam", 0.05), (4, "sgd", 0.1), (8, "adam", 0.05),
                                                (3, "rmsprop", 0.01)])
def test_data_parallel_epoch_driver_all_reduces_between_virtual_ranks(hip_device, world, optimizer, lr):
    """``hiprec_mf_bpr_dp_epoch_fused_range`` with world > 1: every step's fused launch and the all-reduce of its
    [loss partials | gradient] are enqueued by ONE C call that invokes the injected ncclAllReduce itself; gradients
    are scaled by 1 / (batch x world).  Replicas stay bit-identical and follow the oracle on the concatenated global
    batches; the epoch sums are the global ones."""
    U, I, D, bs, epochs = 300, 200, 64, 128, 2
    n_local = 3 * bs + 40
    w0 = onp.init_params(U, I, D, seed=9)
    vw = VirtualWorld(world)
    try:
        res = vw.run(lambda g: dp_rank(g, w0, U, I, D, n_local, bs, optimizer, lr, epochs))
        counters = vw.counters()
    finally:
        vw.close()
    assert all(r["direct"] for r in res)
    steps = (n_local + bs - 1) // bs
    assert counters["all_reduces"] >= epochs * steps, counters
    for r in res[1:]:
        for k in KEYS:
            assert np.array_equal(r["w"][k], res[0]["w"][k]), f"replicas diverged in {k}"
    batches = [tuple(np.concatenate([r["local"][j][k:k + bs] for r in res]) for j in range(3))
               for _ in range(epochs) for k in range(0, n_local, bs)]
    w = onp.copy_params(w0)
    st = onp.new_opt_state(w, optimizer)
    for e in range(epochs):
        tot_loss = tot_reg = 0.0
        for batch in batches[e * steps:(e + 1) * steps]:
            loss, reg = onp.mf_train_step(w, st, batch, "bpr", optimizer, lr)
            tot_loss += loss
            tot_reg += reg
        for r in res:
            assert_scalar_close(r["sums"][e][0], tot_loss, REL, f"epoch {e} loss sum")
            assert_scalar_close(r["sums"][e][1], tot_reg, REL, f"epoch {e} regularizer sum")
    assert_mf_end_state(res[0]["w"], w0, batches, optimizer, lr, f"data-parallel epochs on {world} virtual ranks")


def test_direct_all_reduce_between_virtual_ranks(hip_device):
    """``DirectAllReduce`` (the flat engines' collective): a communicator per rank, in-place fp32 sums of buffers of
    changing sizes, identical bits on every rank."""
    from beta_recsys_amd.replicated import DirectAllReduce

    world = 4

    def rank_fn(group):
        ar = DirectAllReduce(group, torch.device("cuda:0"))
        assert ar.comm is not None
        outs = []
        for n in (5, 100_000, 7, 1_000_003):
            g = torch.Generator(device="cuda").manual_seed(100 * group.rank() + n)
            buf = torch.randn(n, device="cuda", generator=g)
            mine = buf.clone()
            ar.sum_(buf)
            outs.append((mine.cpu(), buf.cpu()))
        return outs

    vw = VirtualWorld(world)
    try:
        res = vw.run(rank_fn)
    finally:
        vw.close()
    for i in range(4):
        want = res[0][i][0].clone()
        for r in res[1:]:
            want += r[i][0]                       # rank order, as the loopback sums
        for r in res:
            assert torch.equal(r[i][1], want)


@pytest.mark.parametrize("routing", ["padded", "variable"])
def test_per_step_sharded_routing_between_virtual_ranks(hip_device, routing):
    """``ShardedMFEngine.train_single_batch`` (A2A-1 triples -> owner(user), A2A-2 ids -> owners and rows back, the
    gradient kernel on fetched rows, A2A-3 gradients -> owners, 3-float all-reduce, local sweep) on 4 virtual ranks
    with the real kernels."""
    from beta_recsys_amd.sharded import ShardedMFEngine

    world, U, I, D, b = 4, 301, 203, 64, 128
    w0 = onp.init_params(U, I, D, seed=3)
    rng = np.random.default_rng(0)
    batches = []
    for _ in range(3):
        batch = (rng.integers(0, U, world * b), zipf(rng, I, world * b), rng.integers(0, I, world * b))
        batches.append(batch)

    def rank_fn(group):
        r = group.rank()
        cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer="adam", lr=0.05,
                             batch_size=b, loss="bpr", routing=routing, sgd_mode="dense"), "system": RUN_DIR}
        with contextlib.redirect_stdout(io.StringIO()):
            eng = ShardedMFEngine(cfg, process_group=group, full_state={k: torch.from_numpy(v) for k, v in w0.items()})
        out = [eng.train_single_batch(tuple(torch.from_numpy(a[r * b:(r + 1) * b]) for a in batch)) for batch in batches]
        full = eng.gather_full_state_dict()
        return out, {k: v.cpu().numpy() for k, v in full.items()}

    vw = VirtualWorld(world)
    try:
        res = vw.run(rank_fn)
    finally:
        vw.close()
    w = onp.copy_params(w0)
    st = onp.new_opt_state(w, "adam")
    for i, batch in enumerate(batches):
        ref_loss, ref_reg = onp.mf_train_step(w, st, batch, "bpr", "adam", 0.05)
        for out, _ in res:
            assert_scalar_close(out[i][0], ref_loss, REL, "loss")
            assert_scalar_close(out[i][1], ref_reg, REL, "reg")
    assert_mf_end_state(res[0][1], w0, batches, "adam", 0.05, f"{routing} routing on {world} virtual ranks")


@pytest.mark.parametrize("kind", ["neumf", "mlp"])
def test_sharded_ncf_tables_between_virtual_ranks(hip_device, kind):
    """``sharded_ncf.py`` on 2 virtual ranks with the real kernels: tables row-sharded, tower replicated; uneven
    splits and an empty rank; losses and the gathered state_dict follow oracle/ncf_numpy.py."""
    import beta_recsys_amd as hp
    from beta_recsys_amd import sharded_ncf
    from oracle import ncf_numpy as onc

    world, U, I, L, emb = 2, 301, 203, 3, 16
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=emb, dropout=0.0, device_str="cuda:0", optimizer="adam", lr=0.01,
                         batch_size=256, model="ncf_end", mlp_config={"n_layers": L}, gmf_config={}), "system": RUN_DIR}
    torch.manual_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        full = getattr(hp, {"neumf": "NeuMF", "mlp": "MLP"}[kind])(dict(cfg["model"], device_str="cpu")).state_dict()
    cls = {"neumf": sharded_ncf.ShardedNeuMFEngine, "mlp": sharded_ncf.ShardedMLPEngine}[kind]
    splits = [(128, 128), (200, 56), (256, 0)]
    rng = np.random.default_rng(1)
    batches = []
    for split in splits:
        B = sum(split)
        users, items = rng.integers(0, U, B), rng.integers(0, I, B)
        items[: B // 4] = items[0]
        batches.append((users, items, (rng.random(B) < 0.2).astype(np.float32)))

    def rank_fn(group):
        r = group.rank()
        eng = cls(cfg, process_group=group, full_state=full)
        losses = []
        for split, (users, items, ratings) in zip(splits, batches):
            lo = sum(split[:r])
            sl = slice(lo, lo + split[r])
            losses.append(eng.train_single_batch(users[sl], items[sl], ratings[sl]))
        return losses, {k: v.cpu().numpy() for k, v in eng.gather_full_state_dict().items()}

    vw = VirtualWorld(world)
    try:
        res = vw.run(rank_fn)
    finally:
        vw.close()
    w = {k: v.numpy().copy() for k, v in full.items()}
    w_start = {k: v.copy() for k, v in w.items()}
    st = onc.new_opt_state(w, "adam")
    for i, batch in enumerate(batches):
        ref = onc.ncf_train_step(w, st, batch, kind, "adam", 0.01)
        for losses, _ in res:
            assert_scalar_close(losses[i], ref, REL, "loss")
    assert_ncf_end_state(res[0][1], w_start, batches, kind, "adam", 0.01, f"sharded {kind} on 2 virtual ranks")
