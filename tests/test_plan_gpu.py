"""The row-sharded engine's epoch planner (csrc/plan.hip) and the planned step's fused launches (csrc/shard.hip)
on the GPU, through the C ABI, against their numpy statements (tests/plan_statements.py): integer work, so the
comparisons are exact wherever the contract fixes an order and property checks where it leaves one open."""
import numpy as np
import pytest
import torch

import plan_statements as ps

pytestmark = pytest.mark.gpu


def kernels(hip_device):
    from beta_recsys_amd.sharded import HipKernels

    return HipKernels(torch.device("cuda:0"))


def zipf_triples(rng, n, U, I):
    p = 1.0 / np.arange(1, I + 1)
    return rng.integers(0, U, n), rng.permutation(I)[rng.choice(I, n, p=p / p.sum())], rng.integers(0, I, n)


@pytest.mark.parametrize("n,bs,R,shuffle", [(5000, 1024, 4, False), (70000, 2500, 8, True), (300, 64, 3, True),
                                             (200_000, 65536, 8, True), (1000, 1000, 1, False), (4097, 4096, 64, False)])
def test_plan_route_is_a_stable_counting_sort(hip_device, n, bs, R, shuffle):
    k = kernels(hip_device)
    rng = np.random.default_rng(n)
    U, I = 10_007, 1_003
    users, pos, neg = zipf_triples(rng, n, U, I)
    perm = rng.permutation(n) if shuffle else None
    dev = lambda a: None if a is None else torch.from_numpy(np.asarray(a, dtype=np.int64)).cuda()  # noqa: E731
    send, cnt = k.plan_route(dev(users), dev(pos), dev(neg), dev(perm), bs, R, U, I)
    ref_send, ref_cnt, status = ps.plan_route(users, pos, neg, perm, bs, R, U, I)
    assert status == 0 and int(k.plan_status()[0]) == 0
    assert np.array_equal(cnt.cpu().numpy(), ref_cnt)
    assert np.array_equal(send.cpu().numpy(), ref_send)


def test_plan_route_flags_and_drops_out_of_range_ids(hip_device):
    """an out-of-range id must not alias into another row's routing; it raises the status bits the host
    turns into IndexError (nn.Embedding's behaviour) and the triple is dropped from the plan."""
    k = kernels(hip_device)
    rng = np.random.default_rng(0)
    n, bs, R, U, I = 3000, 512, 4, 500, 300
    users, pos, neg = zipf_triples(rng, n, U, I)
    users[17], pos[900], neg[2999], users[5] = U, -1, I + 7, -3
    dev = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    send, cnt = k.plan_route(dev(users), dev(pos), dev(neg), None, bs, R, U, I)
    ref_send, ref_cnt, status = ps.plan_route(users, pos, neg, None, bs, R, U, I)
    assert status == 3 and int(k.plan_status()[0]) == 3
    assert np.array_equal(cnt.cpu().numpy(), ref_cnt) and int(ref_cnt.sum()) == n - 4
    assert np.array_equal(send.cpu().numpy()[: n - 4], ref_send)
    with pytest.raises(IndexError):
        k.check_status()
    assert int(k.plan_status()[0]) == 0


@pytest.mark.parametrize("R,S,mean", [(1, 5, 200), (4, 7, 60), (8, 50, 300), (3, 1, 10)])
def test_plan_place_triples_and_requests(hip_device, R, S, mean):
    k = kernels(hip_device)
    rng = np.random.default_rng(R * 100 + S)
    cnt = rng.poisson(mean, (R, S)).astype(np.int32)
    cnt[rng.random((R, S)) < 0.15] = 0          # empty groups
    n = int(cnt.sum())
    recv = rng.integers(0, 1 << 20, (n, 3)).astype(np.int32)
    cap = int(cnt.sum(0).max()) + 3
    U, P, N, fill = k.plan_place_triples(torch.from_numpy(recv).cuda(), torch.from_numpy(cnt).cuda(), S, cap)
    rU, rP, rN = ps.plan_place_triples(recv, cnt, S, cap)
    for got, ref in ((U, rU), (P, rP), (N, rN)):
        assert np.array_equal(got.cpu().numpy(), ref)
    assert np.array_equal(fill.cpu().numpy(), cnt.sum(0))
    # requests: every source's list is free of duplicates, rows repeat ACROSS sources (hot items)
    n_local = 5000
    incoming = np.concatenate([rng.choice(n_local, int(c), replace=False) if c <= n_local else rng.integers(0, n_local, int(c))
                               for c in cnt.reshape(-1)] + [np.zeros(0, dtype=np.int64)]).astype(np.int32)
    in_idx, ex_in, dup = k.plan_place_requests(torch.from_numpy(incoming).cuda(), torch.from_numpy(cnt).cuda(), S, n_local)
    r_idx, r_ex = ps.plan_place_requests(incoming, cnt, S)
    assert np.array_equal(in_idx.cpu().numpy(), r_idx) and np.array_equal(ex_in.cpu().numpy(), r_ex)
    # the duplicate bits: a row more than one peer asks for in a step
    step_len = cnt.sum(0).astype(np.int64) + R
    off = np.concatenate([[0], np.cumsum(step_len)])
    dup = dup.cpu().numpy().view(np.uint32)
    for s in range(S):
        rows = r_idx[off[s]:off[s + 1]]
        rows = rows[rows >= 0]
        want = np.zeros(n_local, dtype=bool)
        vals, c = np.unique(rows, return_counts=True)
        want[vals[c > 1]] = True
        got = ((dup[s][np.arange(n_local) >> 5] >> (np.arange(n_local) & 31)) & 1).astype(bool)
        assert np.array_equal(got, want), f"step {s}: duplicate bits"


@pytest.mark.parametrize("R,S,cap,I", [(1, 3, 500, 97), (4, 5, 700, 211), (8, 4, 3000, 5003), (3, 2, 40, 7),
                                        (8, 3, 20000, 1003)])
def test_plan_item_slots_properties(hip_device, R, S, cap, I):
    """De-duplicated item slots, request lists, grouping by positive item and row ownership of every step -- any
    numbering / order the contract leaves open is accepted, everything it fixes is checked (check_item_slots)."""
    k = kernels(hip_device)
    rng = np.random.default_rng(cap)
    n_users_local = 1 + cap // 3          # many repeated user rows
    U = np.full(S * cap, -1, dtype=np.int64)
    P, N = np.zeros(S * cap, dtype=np.int64), np.zeros(S * cap, dtype=np.int64)
    for s in range(S):
        live = cap - int(rng.integers(0, cap // 4 + 1))
        u, p, q = zipf_triples(rng, live, n_users_local, I)
        U[s * cap: s * cap + live], P[s * cap: s * cap + live], N[s * cap: s * cap + live] = u, p, q
    dev = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    out = k.plan_item_slots(dev(U), dev(P), dev(N), S, cap, R, n_users_local, I)
    torch.cuda.synchronize()
    ps.check_item_slots(out, U, P, N, S, cap, R, n_users_local)
    # with the triples per step given, only the padding behind them is written (no memset of the outputs)
    live = torch.from_numpy((U.reshape(S, cap) >= 0).sum(1).astype(np.int32)).cuda()
    out = k.plan_item_slots(dev(U), dev(P), dev(N), S, cap, R, n_users_local, I, live)
    torch.cuda.synchronize()
    ps.check_item_slots(out, U, P, N, S, cap, R, n_users_local)
    # the statement's own answer satisfies the same contract
    ps.check_item_slots(ps.plan_item_slots(U, P, N, S, cap, R, n_users_local), U, P, N, S, cap, R, n_users_local)


def test_payload_zero_and_apply_finish_with_a_self_segment(hip_device):
    """The two fused launches around the gradient kernel, incl. the rows a rank asks of itself (they bypass the
    exchange buffers): payload + clear, then apply + the step's bookkeeping."""
    from beta_recsys_amd import _lib

    k = kernels(hip_device)
    lib = k.lib
    rng = np.random.default_rng(3)
    n_rows, D, n, n_src = 1000, 24, 700, 3
    ld = D + 1
    emb = torch.from_numpy(rng.standard_normal((n_rows, D)).astype(np.float32)).cuda()
    bias = torch.from_numpy(rng.standard_normal((n_rows, 1)).astype(np.float32)).cuda()
    idx_np = rng.integers(0, n_rows, n).astype(np.int32)
    extra_pos = np.array([199, 449, 699], dtype=np.int32)       # one extra row closes every source's segment
    idx_np[extra_pos] = -1
    idx = torch.from_numpy(idx_np).cuda()
    lo, hi = 200, 450                                           # source 1 is this rank itself
    payload = torch.full((n, ld), 7.0, device="cuda")
    self_dst = torch.full((hi - lo, ld), 7.0, device="cuda")
    zero = torch.full((1234,), 5.0, device="cuda")
    _lib.check(lib.hiprec_shard_payload_zero(_lib.ptr(emb), _lib.ptr(bias), n_rows, D, _lib.ptr(idx), n, lo, hi,
                                             _lib.ptr(payload), _lib.ptr(self_dst), _lib.ptr(zero), zero.numel(), None,
                                             _lib.ptr(k.stats), k._st()))
    ref = np.concatenate([emb.cpu().numpy(), bias.cpu().numpy()], 1)[np.maximum(idx_np, 0)]
    ref[idx_np < 0] = 0.0
    got = payload.cpu().numpy()
    assert np.array_equal(got[:lo], ref[:lo]) and np.array_equal(got[hi:], ref[hi:])
    assert (got[lo:hi] == 7.0).all(), "the self segment must not be written into the exchange buffer"
    assert np.array_equal(self_dst.cpu().numpy(), ref[lo:hi])
    assert float(zero.abs().max()) == 0.0
    # with flags, only the flagged [D + 1]-float rows of the gradient buffer are cleared
    gbuf = torch.full((40, ld), 5.0, device="cuda")
    flags = torch.from_numpy((rng.random(40) < 0.3).astype(np.uint8)).cuda()
    _lib.check(lib.hiprec_shard_payload_zero(_lib.ptr(emb), _lib.ptr(bias), n_rows, D, _lib.ptr(idx), n, lo, hi,
                                             _lib.ptr(payload), _lib.ptr(self_dst), _lib.ptr(gbuf), gbuf.numel(),
                                             _lib.ptr(flags), _lib.ptr(k.stats), k._st()))
    assert torch.equal((gbuf == 0).all(1), flags.bool()) and torch.equal((gbuf == 5.0).all(1), ~flags.bool())

    g_np = rng.standard_normal((n, ld)).astype(np.float32)
    g_np[extra_pos, :3] = [[0.5, 0.25, 0.125], [1.0, 2.0, 4.0], [0.0625, 8.0, 16.0]]
    g_recv = torch.from_numpy(g_np).cuda()
    g_self = g_recv[lo:hi].clone()
    g_recv[lo:hi] = 99.0                                        # must be read from g_self instead
    w_emb, w_bias, scalar = emb.clone(), bias.clone(), torch.tensor([3.0], device="cuda")
    k.reset_clock(0.9, 0.999)
    ep = torch.from_numpy(extra_pos).cuda()
    # duplicate bits as the plan would set them: rows named more than once take the atomic path, the others a plain RMW
    vals, cnts = np.unique(idx_np[idx_np >= 0], return_counts=True)
    bits = np.zeros((n_rows + 31) // 32, dtype=np.uint32)
    for r in vals[cnts > 1]:
        bits[r >> 5] |= np.uint32(1) << np.uint32(r & 31)
    assert (cnts > 1).any() and (cnts == 1).any()
    dup_bits = torch.from_numpy(bits.view(np.int32)).cuda()
    _lib.check(lib.hiprec_shard_apply_finish(_lib.ptr(w_emb), _lib.ptr(w_bias), n_rows, D, _lib.ptr(idx),
                                             _lib.ptr(g_recv), n, lo, hi, _lib.ptr(g_self), -0.5, _lib.ptr(ep), n_src,
                                             _lib.ptr(scalar), -0.5, 1, _lib.ptr(dup_bits), _lib.ptr(k.stats), k._st()))
    ref_emb, ref_bias = emb.cpu().numpy().astype(np.float64), bias.cpu().numpy().astype(np.float64)
    keep = idx_np >= 0
    np.add.at(ref_emb, idx_np[keep], -0.5 * g_np[keep, :D].astype(np.float64))
    np.add.at(ref_bias, idx_np[keep], -0.5 * g_np[keep, D:].astype(np.float64))
    assert np.abs(w_emb.cpu().numpy() - ref_emb).max() < 1e-5 and np.abs(w_bias.cpu().numpy() - ref_bias).max() < 1e-5
    loss, reg, loss_sum, reg_sum = k.epoch_stats()
    assert (loss, reg) == (0.5 + 1.0 + 0.0625, 0.25 + 2.0 + 8.0) and (loss_sum, reg_sum) == (loss, reg)
    assert float(scalar[0]) == 3.0 - 0.5 * (0.125 + 4.0 + 16.0)
    from beta_recsys_amd.mf import read_stats
    assert read_stats(k.stats).step == 1
