"""The loopback world of tests/loopback.py on CPU: its python collectives against their definitions, the native
stand-in's exports, and the row-sharded planned epoch on 4 thread-ranks through the ``beta_recsys_amd._dist`` seam
with the oracle standing in for the kernels (the same check the gloo processes pass) -- so that what the GPU tests
of tests/test_virtual_ranks_gpu.py rely on is itself tested where no GPU is needed."""
import contextlib
import ctypes
import io

import numpy as np
import pytest
import torch

import loopback
from loopback import VirtualWorld
from oracle import mf_numpy as onp


def test_native_stand_in_exports_the_rccl_shaped_entry_points():
    lib = ctypes.CDLL(loopback.LIB_PATH)
    for name in loopback.EXPORTS:
        assert hasattr(lib, name), f"{name} is not exported by {loopback.LIB_PATH}"


def test_python_collectives_between_thread_ranks():
    import beta_recsys_amd._dist as dist

    R = 4

    def rank_fn(group):
        r = dist.get_rank(group)
        assert dist.get_world_size(group) == R and dist.get_backend(group) == "gloo"
        out = {}
        t = torch.tensor([r + 1.0, 10.0 * r])
        dist.all_reduce(t, group=group)
        out["sum"] = t.clone()
        t = torch.tensor([r, -r])
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        out["max"] = t.clone()
        t = torch.tensor([r + 5])
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        out["min"] = t.clone()
        # rank r sends q + 1 rows [100 r + q] to rank q
        send = torch.cat([torch.full((q + 1, 2), 100 * r + q) for q in range(R)])
        recv = torch.empty(((r + 1) * R, 2), dtype=send.dtype)
        dist.all_to_all_single(recv, send, output_split_sizes=[r + 1] * R, input_split_sizes=[q + 1 for q in range(R)],
                               group=group)
        out["a2a"] = recv
        even = torch.arange(R) + 10 * r
        got = torch.empty(R, dtype=even.dtype)
        dist.all_to_all_single(got, even, group=group)
        out["a2a_even"] = got
        parts = [torch.empty(3) for _ in range(R)]
        dist.all_gather(parts, torch.full((3,), float(r)), group=group)
        out["gather"] = torch.stack(parts)
        b = torch.tensor([float(r)])
        dist.broadcast(b, src=2, group=group)
        out["bcast"] = b
        side = dist.new_group(like=group)
        s = torch.tensor([1.0])
        dist.all_reduce(s, group=side)
        out["side"] = s
        return out

    res = VirtualWorld(R, device="cpu").run(rank_fn)
    for r, out in enumerate(res):
        assert out["sum"].tolist() == [10.0, 60.0] and out["max"].tolist() == [3, 0] and out["min"].tolist() == [5]
        assert out["a2a"].tolist() == [[100 * q + r] * 2 for q in range(R) for _ in range(r + 1)]
        assert out["a2a_even"].tolist() == [10 * q + r for q in range(R)]
        assert out["gather"].tolist() == [[float(q)] * 3 for q in range(R)]
        assert out["bcast"].item() == 2.0 and out["side"].item() == R


def test_mismatched_split_sizes_fail_instead_of_hanging():
    import beta_recsys_amd._dist as dist

    def rank_fn(group):
        r = dist.get_rank(group)
        send = torch.zeros(4)
        recv = torch.zeros(4 if r == 0 else 5)
        dist.all_to_all_single(recv, send, output_split_sizes=[2, 2] if r == 0 else [2, 3], input_split_sizes=[2, 2],
                               group=group)

    with pytest.raises(RuntimeError, match="expects 3 rows from rank 1, which sends 2"):
        VirtualWorld(2, device="cpu", timeout=5.0).run(rank_fn)


def test_a_rank_that_raises_releases_its_peers():
    import beta_recsys_amd._dist as dist

    def rank_fn(group):
        if dist.get_rank(group) == 1:
            raise ValueError("rank 1 gives up")
        dist.all_reduce(torch.zeros(1), group=group)

    with pytest.raises(ValueError, match="rank 1 gives up"):
        VirtualWorld(3, device="cpu", timeout=20.0).run(rank_fn)


@pytest.mark.parametrize("optimizer,lr", [("sgd", 0.1), ("adam", 0.05)])
def test_planned_sharded_epoch_on_thread_ranks_equals_single_process(optimizer, lr):
    """ShardedMFEngine.plan_epoch / run_planned_epoch over the loopback group (4 thread-ranks, oracle kernels): the
    host code reaches its peers through ``_dist`` only, so it runs unchanged over this group -- and passes the check
    the gloo processes pass."""
    import beta_recsys_amd as hp
    from beta_recsys_amd.sharded import ShardedMFEngine
    from test_sharded_gloo import OraclePlannedKernels, check_planned, make_config

    world, n_local, bs, U, I, D = 4, 41, 8, 37, 23, 8
    w0 = onp.init_params(U, I, D, seed=7)

    def rank_fn(group):
        rank = group.rank()
        rng = np.random.default_rng(50 + rank)
        users, neg = rng.integers(0, U, n_local), rng.integers(0, I, n_local)
        p = 1.0 / np.arange(1, I + 1)
        pos = rng.choice(I, n_local, p=p / p.sum())
        with contextlib.redirect_stdout(io.StringIO()):
            eng = ShardedMFEngine(make_config(U, I, D, optimizer, lr, "padded", "rows"), process_group=group,
                                  kernels=OraclePlannedKernels(),
                                  full_state={k: torch.from_numpy(v) for k, v in w0.items()})
        loader = hp.DeviceTripleBatcher(torch.from_numpy(users), torch.from_numpy(pos), torch.from_numpy(neg), bs,
                                        shuffle=True, generator=torch.Generator().manual_seed(9 + rank))
        stats = eng.run_planned_epoch(eng.plan_epoch(loader))
        full = eng.gather_full_state_dict()
        order = torch.randperm(n_local, generator=torch.Generator().manual_seed(9 + rank)).numpy()
        return stats, {k: v.numpy() for k, v in full.items()}, (users[order], pos[order], neg[order])

    res = VirtualWorld(world, device="cpu").run(rank_fn)
    for stats, full, _ in res:
        check_planned({"stats": stats, "full": full, "w0": w0, "local": [r[2] for r in res], "bs": bs}, n_local, bs,
                      optimizer, lr)


def test_an_out_of_range_id_on_one_rank_raises_on_every_rank():
    """the plan's status word is OR-ed over the group before the host looks at it -- every rank raises
    IndexError between the same two collectives (a rank-local raise left the peers hanging in the next all-to-all),
    the status is cleared and the engines plan the corrected epoch afterwards."""
    import beta_recsys_amd as hp
    from beta_recsys_amd.sharded import ShardedMFEngine
    from test_sharded_gloo import OraclePlannedKernels, make_config

    world, n_local, bs, U, I, D = 3, 20, 8, 37, 23, 8
    w0 = onp.init_params(U, I, D, seed=7)

    def rank_fn(group):
        rank = group.rank()
        rng = np.random.default_rng(50 + rank)
        users, pos, neg = rng.integers(0, U, n_local), rng.integers(0, I, n_local), rng.integers(0, I, n_local)
        with contextlib.redirect_stdout(io.StringIO()):
            eng = ShardedMFEngine(make_config(U, I, D, "sgd", 0.1, "padded", "rows"), process_group=group,
                                  kernels=OraclePlannedKernels(),
                                  full_state={k: torch.from_numpy(v) for k, v in w0.items()})
        bad = pos.copy()
        if rank == 1:
            bad[5] = I          # only rank 1 holds the bad id
        loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a) for a in (users, bad, neg)), bs, shuffle=False)
        with pytest.raises(IndexError, match="item index"):
            eng.plan_epoch(loader)
        loader = hp.DeviceTripleBatcher(*(torch.from_numpy(a) for a in (users, pos, neg)), bs, shuffle=False)
        return eng.run_planned_epoch(eng.plan_epoch(loader))

    res = VirtualWorld(world, device="cpu", timeout=20.0).run(rank_fn)
    assert all(np.isfinite(r[2]) for r in res)
