"""Plain numpy / torch statements of the row-sharded engine's epoch-planner kernels (csrc/plan.hip) and of the
planned step's fused launches (csrc/shard.hip).  TEST INFRASTRUCTURE: the gloo tests run the engine's host logic
on CPU with these standing in for the HIP kernels (the ``kernels=`` seam of ShardedMFEngine), and the ``-m gpu``
tests check the kernels against them.  The reference has no counterpart (it is single-device); the semantics they
serve are beta_rec/models/mf.py:92-119 on the concatenated global batch.

Contracts (R = world, S = steps of the epoch, cap = triples per step block):
  plan_route           exact: stable order inside a (destination, step) group
  plan_place_triples   exact
  plan_item_slots      any numbering of a (step, owner) chunk's items and any order inside a group of equal positive
                       items is legal; `check_item_slots` states what must hold
  plan_place_requests  exact
"""
import numpy as np
import torch


def plan_route(users, pos, neg, perm, bs, R, n_users, n_items):
    """-> (send int32 [n_ok, 3] = (user // R, pos, neg) ordered by (user % R, step, visiting order),
    cnt_ds int32 [R, S], status bits)."""
    users, pos, neg = (np.asarray(a, dtype=np.int64) for a in (users, pos, neg))
    n = len(users)
    S = max((n + bs - 1) // bs, 1)
    idx = np.arange(n) if perm is None else np.asarray(perm, dtype=np.int64)
    u, p, q = users[idx], pos[idx], neg[idx]
    u_ok = (u >= 0) & (u < n_users)
    i_ok = (p >= 0) & (p < n_items) & (q >= 0) & (q < n_items)
    ok = u_ok & i_ok
    status = (0 if u_ok.all() else 1) | (0 if i_ok.all() else 2)
    step = np.arange(n) // bs
    key = (u % R) * S + step
    order = np.argsort(np.where(ok, key, R * S), kind="stable")[: int(ok.sum())]
    send = np.stack([u // R, p, q], 1)[order].astype(np.int32)
    cnt_ds = np.bincount(key[ok], minlength=R * S).reshape(R, S).astype(np.int32)
    return send, cnt_ds, status


def plan_place_triples(recv, recv_cnt, S, cap):
    """recv int32 [n_recv, 3] (source, step)-ordered; recv_cnt [R, S] -> U, P, N int64 [S * cap]."""
    recv, recv_cnt = np.asarray(recv, dtype=np.int64).reshape(-1, 3), np.asarray(recv_cnt, dtype=np.int64)
    R = recv_cnt.shape[0]
    U = np.full(S * cap, -1, dtype=np.int64)
    P, N = np.zeros(S * cap, dtype=np.int64), np.zeros(S * cap, dtype=np.int64)
    j = 0
    fill = np.zeros(S, dtype=np.int64)
    # destination of group (q, s): after the groups of lower sources in the same step
    within = np.cumsum(recv_cnt, 0) - recv_cnt
    for q in range(R):
        for s in range(S):
            c = int(recv_cnt[q, s])
            at = s * cap + int(within[q, s])
            U[at:at + c], P[at:at + c], N[at:at + c] = recv[j:j + c, 0], recv[j:j + c, 1], recv[j:j + c, 2]
            j += c
            fill[s] += c
    assert j == len(recv) and fill.max(initial=0) <= cap
    return U, P, N


def ownership(U, SP, SN, cap, n_rows_u, n_slots):
    """own int32 [3, S * cap] / total int32 [S, stride] in the convention of csrc/ownership.hip as the planned step
    uses it: a row's `own` is a slot id of its step, total[step][slot] its number of occurrences in the step; user
    rows and item slots use distinct ids (here: user row r -> r, item slot t -> n_rows_u + t)."""
    S = len(U) // cap
    stride = n_rows_u + n_slots
    own = np.full((3, len(U)), -1, dtype=np.int32)
    total = np.zeros((S, stride), dtype=np.int32)
    for s in range(S):
        blk = slice(s * cap, (s + 1) * cap)
        live = U[blk] >= 0
        ids = (U[blk][live], n_rows_u + SP[blk][live], n_rows_u + SN[blk][live])
        for role in range(3):
            np.add.at(total[s], ids[role], 1)
            own[role, blk][live] = ids[role]
    return own, total, stride


def plan_item_slots(U, P, N, S, cap, R, n_users_local):
    """One legal answer: a step's distinct items numbered owner by owner in ascending item order, its triples sorted
    (stably) by positive slot."""
    U, P, N = (np.asarray(a, dtype=np.int64) for a in (U, P, N))
    U2 = np.full(S * cap, -1, dtype=np.int64)
    SP, SN = np.zeros(S * cap, dtype=np.int64), np.zeros(S * cap, dtype=np.int64)
    req_cnt = np.zeros((S, R), dtype=np.int32)
    ex_req = np.zeros((S, R), dtype=np.int32)
    chunks = [[None] * R for _ in range(S)]
    for s in range(S):
        blk = slice(s * cap, (s + 1) * cap)
        live = U[blk] >= 0
        u, p, q = U[blk][live], P[blk][live], N[blk][live]
        items = np.unique(np.concatenate([p, q]))
        slot_of, run = {}, 0
        for d in range(R):
            mine = items[items % R == d]
            chunks[s][d] = mine // R
            for it in mine:
                slot_of[int(it)] = run
                run += 1
            req_cnt[s, d] = len(mine)
            ex_req[s, d] = run
            run += 1
        sp = np.array([slot_of[int(x)] for x in p], dtype=np.int64)
        sn = np.array([slot_of[int(x)] for x in q], dtype=np.int64)
        o = np.argsort(sp, kind="stable")
        c = len(o)
        U2[s * cap: s * cap + c], SP[s * cap: s * cap + c], SN[s * cap: s * cap + c] = u[o], sp[o], sn[o]
    req_send = np.concatenate([chunks[s][d] for d in range(R) for s in range(S)] + [np.zeros(0, dtype=np.int64)])
    n_slots = (req_cnt.sum(1) + R).astype(np.int64)
    own, total, stride = ownership(U2, SP, SN, cap, max(n_users_local, 1), int(n_slots.max()))
    return {"U": U2, "SP": SP, "SN": SN, "own": own, "total": total, "stride": stride, "req_cnt": req_cnt,
            "req_ds": np.ascontiguousarray(req_cnt.T), "ex_req": ex_req, "req_send": req_send.astype(np.int32)}


def check_item_slots(out, U, P, N, S, cap, R, n_users_local):
    """What ANY legal plan_item_slots answer satisfies, given the step blocks it was made from."""
    U, P, N = (np.asarray(a, dtype=np.int64) for a in (U, P, N))
    o = {k: (v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items() if k != "stride"}
    req_cnt = o["req_cnt"].astype(np.int64)
    assert np.array_equal(o["req_ds"], req_cnt.T)
    send_base = np.concatenate([[0], np.cumsum(req_cnt.T.reshape(-1))])   # (destination, step)-ordered request list
    for s in range(S):
        blk = slice(s * cap, (s + 1) * cap)
        live_in = U[blk] >= 0
        n_live = int(live_in.sum())
        u2, sp, sn = o["U"][blk], o["SP"][blk], o["SN"][blk]
        assert (u2[:n_live] >= 0).all() and (u2[n_live:] == -1).all(), "live triples first, padding behind"
        # the slot -> item map the owners will be told: chunk d of the step = req_send[send_base[d, s] ...]
        chunk_start = np.cumsum(req_cnt[s] + 1) - (req_cnt[s] + 1)
        n_slots = int((req_cnt[s] + 1).sum())
        item_of = np.full(n_slots, -1, dtype=np.int64)
        for d in range(R):
            rows = o["req_send"][send_base[d * S + s]: send_base[d * S + s + 1]].astype(np.int64)
            assert len(rows) == req_cnt[s, d]
            item_of[chunk_start[d]: chunk_start[d] + len(rows)] = rows * R + d
            assert o["ex_req"][s, d] == chunk_start[d] + req_cnt[s, d]
        real = item_of[item_of >= 0]
        assert len(np.unique(real)) == len(real), "an item asked for twice in one step"
        want = np.unique(np.concatenate([P[blk][live_in], N[blk][live_in]]))
        assert np.array_equal(np.sort(real), want), "the requests are exactly the step's distinct items"
        # the re-laid block holds the same triples (as a multiset), items expressed as slots
        got = np.stack([u2[:n_live], item_of[sp[:n_live]], item_of[sn[:n_live]]], 1)
        ref = np.stack([U[blk][live_in], P[blk][live_in], N[blk][live_in]], 1)
        assert np.array_equal(got[np.lexsort(got.T[::-1])], ref[np.lexsort(ref.T[::-1])])
        # grouped by positive item: every slot's occurrences are adjacent
        change = np.flatnonzero(np.diff(sp[:n_live]) != 0)
        assert len(change) + (1 if n_live else 0) == len(np.unique(sp[:n_live])), "equal positive items are not adjacent"
        # ownership: occurrences per row of the step
        own, total = o["own"][:, blk], o["total"][s]
        for role, ids in ((0, u2[:n_live]), (1, sp[:n_live]), (2, sn[:n_live])):
            pool = u2[:n_live] if role == 0 else np.concatenate([sp[:n_live], sn[:n_live]])
            vals, cnts = np.unique(pool, return_counts=True)
            cnt_of = dict(zip(vals.tolist(), cnts.tolist()))
            for t in range(n_live):
                slot = int(own[role, t])
                c = cnt_of[int(ids[t])]
                assert (slot < 0 and c == 1) or int(total[slot]) == c, "ownership does not match the batch"
        assert (own[:, n_live:] == -1).all()
        if "slot_shared" in o and o["slot_shared"] is not None:   # 1 = several references (or an extra row)
            flags = o["slot_shared"][s][:n_slots]
            refs = np.bincount(np.concatenate([sp[:n_live], sn[:n_live]]), minlength=n_slots)
            want_flags = (refs > 1).astype(np.uint8)
            want_flags[o["ex_req"][s]] = 1
            assert np.array_equal(flags, want_flags), "slot_shared does not say which slots several triples add into"


def plan_place_requests(incoming, in_qs, S):
    """incoming int32 [n_in] (source, step)-ordered; in_qs [R, S] -> in_idx int32 packed per step with one extra
    row (-1) behind every source's rows, ex_in [S, R]."""
    incoming, in_qs = np.asarray(incoming, dtype=np.int64), np.asarray(in_qs, dtype=np.int64)
    R = in_qs.shape[0]
    step_len = in_qs.sum(0) + R
    step_off = np.concatenate([[0], np.cumsum(step_len)])
    in_idx = np.full(int(step_off[-1]), -1, dtype=np.int32)
    ex_in = np.zeros((S, R), dtype=np.int32)
    j = 0
    for q in range(R):
        for s in range(S):
            c = int(in_qs[q, s])
            at = int(step_off[s]) + int(in_qs[:q, s].sum()) + q
            in_idx[at: at + c] = incoming[j: j + c]
            j += c
    for s in range(S):
        ex_in[s] = np.cumsum(in_qs[:, s]) + np.arange(R)
    assert j == len(incoming)
    return in_idx, ex_in
