"""GPU parity tests of LightGCN: CSR SpMM, propagation, loss/backward and the full step vs golden
vectors from the real reference (same dropped edges for the same torch seed) and vs the
numpy/scipy oracle at BASELINE's C5 shape (ML-1M-sized graph, 3 layers, dim 64, batch 1024)."""
import contextlib
import ctypes
import functools
import io

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from helpers import REL  # noqa: F401  (north_star: 1e-5 relative)
from helpers import EPS32, assert_grads_as_accurate, assert_scalar_close, assert_tensor_close, float64_oracle, load_golden, to64
from oracle import lightgcn_numpy as olg
from test_oracle_golden_lightgcn import golden_adj, golden_mask, params

pytestmark = pytest.mark.gpu


def torch_adj(adj_csr):
    co = adj_csr.tocoo()
    idx = torch.from_numpy(np.vstack((co.row, co.col)).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(co.data.astype(np.float32)), torch.Size(co.shape))


def make_engine(U, I, D, L, optimizer, lr, B, adj_csr, keep=0.6, decay=1e-5, **extra):
    import beta_recsys_amd as hp

    model = dict(n_users=U, n_items=I, emb_dim=D, layer_size=[D] * L, keep_pro=keep, regs=[decay],
                 device_str="cuda:0", optimizer=optimizer, lr=lr, batch_size=B, norm_adj=torch_adj(adj_csr))
    model.update(extra)
    with contextlib.redirect_stdout(io.StringIO()):
        return hp.LightGCNEngine({"model": model, "system": {"run_dir": "/tmp/hiprec_test_runs"}})


def load_weights(eng, w):
    eng.model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})


def get_weights(eng):
    return {k: v.detach().cpu().numpy() for k, v in eng.model.state_dict().items()}


@pytest.mark.parametrize("dim", [8, 64, 100, 200])
def test_spmm_csr_vs_scipy(hip_device, dim):
    from beta_recsys_amd import _lib
    from beta_recsys_amd.lightgcn import _csr_from_coo

    lib = _lib.load()
    rng = np.random.default_rng(dim)
    n, nnz = 700, 9000
    rows = rng.integers(0, n, nnz)
    rows[:3000] = 5  # one very heavy row (spans many edge slices) ; some rows stay empty
    cols = rng.integers(0, n, nnz)
    a = sp.coo_matrix((rng.standard_normal(nnz).astype(np.float32), (rows, cols)), shape=(n, n)).tocsr()
    a.sum_duplicates()
    a.sort_indices()
    co = a.tocoo()
    rp, c, v, _ = _csr_from_coo(torch.from_numpy(co.row.astype(np.int64)), torch.from_numpy(co.col.astype(np.int64)),
                                torch.from_numpy(co.data), n, hip_device)
    from beta_recsys_amd.lightgcn import _slice_rows

    sr = _slice_rows(rp, c, v, n, a.nnz)
    starts = np.arange(len(sr)) * 256
    assert np.array_equal(sr.cpu().numpy(), np.searchsorted(a.indptr, starts, side="right") - 1)
    # with the per-slice first rows (what the engines pass) and without (binary search per wave)
    csr = _lib.Csr(rp.data_ptr(), c.data_ptr(), v.data_ptr(), None, n, a.nnz, _lib.ptr(sr if dim != 100 else None))
    x = rng.standard_normal((n, dim)).astype(np.float32)
    keep = (rng.random(a.nnz) < 0.6)
    xt = torch.from_numpy(x).cuda()
    y = torch.full((n, dim), 7.0, device="cuda")
    acc = torch.ones((n, dim), device="cuda")
    kt = torch.from_numpy(keep.astype(np.uint8)).cuda()
    st = _lib.stream_ptr(hip_device)
    _lib.check(lib.hiprec_spmm_csr(ctypes.byref(csr), _lib.ptr(kt), 1 / 0.6, _lib.ptr(xt), _lib.ptr(y),
                                   _lib.ptr(acc), dim, st))
    dropped = olg.apply_edge_dropout(a, keep, 0.6)
    ref = dropped.astype(np.float64) @ x.astype(np.float64)
    scale = abs(dropped).astype(np.float64) @ np.abs(x).astype(np.float64) + 1e-6
    assert np.all(np.abs(y.cpu().numpy() - ref) <= 64 * EPS32 * scale + 1e-6)
    assert np.all(np.abs(acc.cpu().numpy() - 1.0 - ref) <= 64 * EPS32 * scale + 1e-5)
    _lib.check(lib.hiprec_spmm_csr(ctypes.byref(csr), None, 1.0, _lib.ptr(xt), _lib.ptr(y), None, dim, st))
    ref2 = a.astype(np.float64) @ x.astype(np.float64)
    assert np.all(np.abs(y.cpu().numpy() - ref2) <= 64 * EPS32 * (abs(a).astype(np.float64) @ np.abs(x)) + 1e-6)


def sliced_graph(rp, col, val, eid, n, nnz, dim, n_groups, factor=True, lane_slots=None):
    """hiprec_sliced_csr of a CSR already on the device, as LightGCN.graph() builds it (+ the arrays it points to)."""
    from beta_recsys_amd import _lib
    from beta_recsys_amd.lightgcn import sliced_graph_device, sliced_graph_host

    cap = _lib.load().hiprec_sliced_row_cap(n, dim)
    host = sliced_graph_host(rp.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy(),
                             None if eid is None else eid.cpu().numpy(), n_groups, cap, factor=factor,
                             lane_slots=lane_slots)
    assert host is not None
    sc, hold = sliced_graph_device(host, n, n_groups, cap, "cuda")
    return sc, hold, host


@pytest.mark.parametrize("lane_slots", [None, 16, 24, 32, 48])   # None: the geometry the host picks for the graph
@pytest.mark.parametrize("n,dim,width,n_groups", [(700, 64, 4, 16), (9746, 64, 4, 16), (10100, 100, 4, 3),
                                                  (15001, 32, 2, 7), (19000, 8, 2, 1), (500, 6, 2, 40)])
def test_spmm_sliced_vs_scipy(hip_device, n, dim, width, n_groups, lane_slots):
    """The column-sliced SpMM (source slice in LDS, rows owned by one workgroup): forward graph and transposed graph
    with the forward keep bytes, heavy and empty rows, every accumulate mode, the layout transposes; general values
    (the step's stream = dropped values; lane_slots None) and rank-one values (stream = dropped columns, scaled source
    / result) at every number of slots per lane the kernel is built for."""
    factored = lane_slots is not None
    from beta_recsys_amd import _lib
    from beta_recsys_amd.lightgcn import _csr_from_coo

    lib = _lib.load()
    assert lib.hiprec_sliced_width(n, dim) == width
    assert lib.hiprec_sliced_width(65536, dim) == 0 and lib.hiprec_sliced_width(30000, 64) == 0
    rng = np.random.default_rng(n + dim)
    nnz = 30 * n
    rows = rng.integers(0, n, nnz)
    rows[: nnz // 5] = 5
    rows[rows > n - 40] = n - 41  # the last rows stay empty
    cols = rng.integers(0, n, nnz)
    a = sp.coo_matrix((rng.standard_normal(nnz).astype(np.float32), (rows, cols)), shape=(n, n)).tocsr()
    a.sum_duplicates()
    a.sort_indices()
    if factored:  # val[i][j] = r[i] * c[j], as a degree-normalised adjacency has them
        fr, fc = (rng.random(n).astype(np.float32) + 0.5 for _ in range(2))
        co = a.tocoo()
        a = sp.csr_matrix((fr[co.row] * fc[co.col], (co.row, co.col)), shape=(n, n))
        a.sort_indices()
    co = a.tocoo()
    r64, c64 = torch.from_numpy(co.row.astype(np.int64)), torch.from_numpy(co.col.astype(np.int64))
    rp, c, v, _ = _csr_from_coo(r64, c64, torch.from_numpy(co.data), n, hip_device)
    rpt, ct, vt, order_t = _csr_from_coo(c64, r64, torch.from_numpy(co.data), n, hip_device)
    eid_t = order_t.to(torch.int32).cuda()
    x = rng.standard_normal((n, dim)).astype(np.float32)
    keep = rng.random(a.nnz) < 0.6
    kt = torch.from_numpy(keep.astype(np.uint8)).cuda()
    xt = torch.from_numpy(x).cuda()
    st = _lib.stream_ptr(hip_device)
    xs = torch.full((n * dim,), 3.0, device="cuda")
    _lib.check(lib.hiprec_to_sliced(_lib.ptr(xt), n, dim, width, None, _lib.ptr(xs), st))
    assert np.array_equal(xs.cpu().numpy().reshape(dim // width, n, width),
                          x.reshape(n, dim // width, width).transpose(1, 0, 2))
    back = torch.ones((n, dim), device="cuda")
    _lib.check(lib.hiprec_from_sliced(_lib.ptr(xs), n, dim, width, _lib.ptr(back), 1, st))
    assert np.array_equal(back.cpu().numpy(), x + 1.0)
    dropped = olg.apply_edge_dropout(a, keep, 0.6)
    for graph, (rowptr, col, val, eid) in (("a", (rp, c, v, None)), ("at", (rpt, ct, vt, eid_t))):
        sc, hold, host = sliced_graph(rowptr, col, val, eid, n, a.nnz, dim, n_groups, lane_slots=lane_slots)
        assert ("col_scale" in host) == factored
        S = host["lane_slots"]
        assert S == (lane_slots or 16)
        assert np.diff(host["spill_ptr"]).max() <= lib.hiprec_sliced_row_cap(n, dim)
        lens = np.diff(rowptr.cpu().numpy())
        live_chunks = int((((host["chunks"][:, 1] >> 16) & 0xFF) > 0).sum())  # + empty chunks that pad the subgroups
        assert live_chunks == int(((lens + 4 * S - 1) // (4 * S)).sum())
        assert host["pad_slot"] == int(((lens + S - 1) // S * S).sum())   # then the all-padding tail
        assert host["pad_slot"] + S <= host["n_slots"] < host["pad_slot"] + S + 16 and host["n_slots"] % 16 == 0
        assert host["n_chunks"] % 16 == 0 and host["n_chunks"] >= live_chunks
        step_edges = torch.full((host["n_slots"],), 9.0, device="cuda")
        _lib.check(lib.hiprec_sliced_drop_values(ctypes.byref(sc), _lib.ptr(kt), _lib.ptr(step_edges), st))
        cs = hold.get("col_scale")  # a factored graph takes col_scale (.) X and returns col_scale (.) Y
        src = torch.empty_like(xs)
        _lib.check(lib.hiprec_to_sliced(_lib.ptr(xt), n, dim, width, _lib.ptr(cs), _lib.ptr(src), st))
        unscale = 1.0 if cs is None else 1.0 / cs.cpu().numpy().astype(np.float64)[:, None]
        m = dropped if graph == "a" else dropped.T.tocsr()
        ref = m.astype(np.float64) @ x.astype(np.float64)
        tol = 64 * EPS32 * (abs(m).astype(np.float64) @ np.abs(x).astype(np.float64)) + 1e-6
        ys = torch.full((n * dim,), 7.0, device="cuda")  # stale contents: every row is written by the kernel
        accs = xs.clone()
        _lib.check(lib.hiprec_spmm_sliced(ctypes.byref(sc), _lib.ptr(step_edges), 1 / 0.6, _lib.ptr(src),
                                          _lib.ptr(ys), _lib.ptr(accs), 1, dim, width, st))
        y = torch.empty((n, dim), device="cuda")
        _lib.check(lib.hiprec_from_sliced(_lib.ptr(ys), n, dim, width, _lib.ptr(y), 0, st))
        assert np.all(np.abs(y.cpu().numpy() * unscale - ref) <= tol), graph
        if graph == "a":
            assert np.all(y.cpu().numpy()[n - 39:] == 0), "empty rows are written as zeros"
        _lib.check(lib.hiprec_from_sliced(_lib.ptr(accs), n, dim, width, _lib.ptr(y), 0, st))
        assert np.all(np.abs(y.cpu().numpy() - x - ref) <= tol + 1e-6), graph
        # no dropout, accs = Y
        m2 = a if graph == "a" else a.T.tocsr()
        _lib.check(lib.hiprec_spmm_sliced(ctypes.byref(sc), None, 1.0, _lib.ptr(src), _lib.ptr(ys), _lib.ptr(accs),
                                          2, dim, width, st))
        if not factored:
            assert torch.equal(ys, accs)
        _lib.check(lib.hiprec_from_sliced(_lib.ptr(accs), n, dim, width, _lib.ptr(y), 0, st))
        ref2 = m2.astype(np.float64) @ x.astype(np.float64)
        tol2 = 64 * EPS32 * (abs(m2).astype(np.float64) @ np.abs(x)) + 1e-6
        assert np.all(np.abs(y.cpu().numpy() - ref2) <= tol2)
        _lib.check(lib.hiprec_from_sliced(_lib.ptr(ys), n, dim, width, _lib.ptr(y), 0, st))
        assert np.all(np.abs(y.cpu().numpy() * unscale - ref2) <= tol2)
    with pytest.raises(RuntimeError, match="slice width"):
        _lib.check(lib.hiprec_spmm_sliced(ctypes.byref(sc), None, 1.0, _lib.ptr(xs), _lib.ptr(ys), None, 0, dim,
                                          width * 2 if width < 4 else 2, st))


@pytest.mark.parametrize("spmm", ["auto", "sliced_values", "gather"])
@pytest.mark.parametrize("case", ["lightgcn_adam", "lightgcn_sgd_d64"])
def test_lightgcn_step_matches_reference(hip_device, case, spmm):
    g = load_golden(case)
    U, I, D, L, B, n_steps, seed = (int(x) for x in g["meta"])
    opt, lr, keep, decay = str(g["optimizer"]), float(g["lr"]), float(g["keep"]), float(g["decay"])
    adj = golden_adj(g)
    torch.manual_seed(seed)
    eng = make_engine(U, I, D, L, opt, lr, B, adj, keep, decay, spmm=spmm)
    assert (eng.model.graph()["slice_w"] > 0) == (spmm != "gather")
    if spmm != "gather":  # the reference's D^-1 (A + I) has rank-one values: 2-byte edge stream
        assert ("col_scale" in eng.model.graph()["sliced"][1]) == (spmm == "auto")
    w_init = get_weights(eng)
    for k in w_init:  # same seed -> the reference's xavier init, bit for bit
        assert np.array_equal(w_init[k], g[f"w0/{k}"]), k
    assert list(w_init.keys()) == ["user_embedding.weight", "item_embedding.weight"]
    for s in range(n_steps):
        batch = tuple(torch.from_numpy(g[k][s]) for k in ("users", "pos", "neg"))
        load_weights(eng, params(g, f"w{s}"))
        torch.manual_seed(1000 + s)
        loss, grads = eng.backward_only(batch)
        kept = eng.model.last_keep_mask().cpu().numpy().astype(bool)
        assert np.array_equal(kept, golden_mask(g, s)), "same seed must drop the same edges"
        assert_scalar_close(loss, g["losses"][s], what=f"loss step {s}")
        with float64_oracle(olg):
            _, exact = olg.lightgcn_grads(to64(params(g, f"w{s}")), olg.apply_edge_dropout(adj, golden_mask(g, s), keep).astype(np.float64),
                                          L, *(g[k][s] for k in ("users", "pos", "neg")), decay)
        assert_grads_as_accurate(grads, {k: g[f"g{s + 1}/{k}"] for k in olg.KEYS}, exact, f"grad step {s}")
        if opt == "sgd":
            torch.manual_seed(1000 + s)
            loss2 = eng.train_single_batch(batch)
            assert_scalar_close(loss2, g["losses"][s], what=f"loss (step) {s}")
            w = get_weights(eng)
            for k in olg.KEYS:
                assert_tensor_close(w[k], g[f"w{s + 1}/{k}"], 1e-6, f"weights {k} step {s}")
    if opt == "adam":  # chained trajectory from w0 with the reference's seeds
        load_weights(eng, params(g, "w0"))
        eng.load_optimizer_state(0)
        for s in range(n_steps):
            torch.manual_seed(1000 + s)
            loss = eng.train_single_batch(tuple(torch.from_numpy(g[k][s]) for k in ("users", "pos", "neg")))
            assert_scalar_close(loss, g["losses"][s], REL, f"trajectory loss {s}")
        # every element on the reference's trajectory (helpers.oracle_trajectory: oracle runs with the reference's
        # dropped edges and every gradient moved by 1e-5 of its scale give the legal envelope)
        from helpers import assert_on_trajectory, oracle_trajectory

        steps_np = [(golden_mask(g, s), tuple(g[k][s] for k in ("users", "pos", "neg"))) for s in range(n_steps)]
        _, env, upd = oracle_trajectory(
            params(g, "w0"), steps_np,
            lambda w, b: olg.lightgcn_grads(w, olg.apply_edge_dropout(adj, b[0], keep), L, b[1][0], b[1][1], b[1][2],
                                            decay)[1],
            lambda w, gr, st: olg.opt_step(w, gr, st, "adam", lr), lambda w: olg.new_opt_state(w, "adam"))
        assert_on_trajectory(get_weights(eng), params(g, f"w{n_steps}"), env, upd, "LightGCN trajectory")
    load_weights(eng, params(g, f"w{n_steps}"))
    scores = eng.model.predict(g["probe_users"], g["probe_items"])
    assert_tensor_close(scores.cpu().numpy(), g["probe_scores"], 1e-5, "probe scores")
    assert not eng.model.training
    ue, ie = eng.model.forward()  # eval mode: no dropout
    ref_u, ref_i = olg.propagate(params(g, f"w{n_steps}"), adj, L)
    assert_tensor_close(ue.cpu().numpy(), ref_u, 1e-5, "propagated users")
    assert_tensor_close(ie.cpu().numpy(), ref_i, 1e-5, "propagated items")


@functools.lru_cache(maxsize=2)
def ml1m_like_graph(seed=0, U=6040, I=3706, n_edges=988_000):
    """a bipartite graph with ML-1M's degree profile and 988 k UNIQUE train edges -> nnz(D^-1 (A + I)) =
    2 x 988 000 + 9 746 = 1.99 M (rounds 1-2 drew 1 M edges WITH duplicates: 745 k unique, nnz 1.49 M).  Zipf items,
    duplicates redrawn until the count is met."""
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, I + 1)
    p /= p.sum()
    item_of = rng.permutation(I)
    pairs = np.zeros(0, dtype=np.int64)
    while pairs.size < n_edges:
        m = int((n_edges - pairs.size) * 1.3) + 1000
        new = rng.integers(0, U, m) * I + item_of[rng.choice(I, m, p=p)]
        pairs = np.unique(np.concatenate([pairs, new]))
    pairs = rng.permutation(pairs)[:n_edges]
    return olg.build_norm_adj(U, I, pairs // I, pairs % I)


@pytest.mark.parametrize("spmm", ["auto", "sliced_values", "gather"])
def test_lightgcn_full_size_c5_vs_oracle(hip_device, spmm):
    """BASELINE configs[4] shape: 988 k unique interactions (nnz 1.99 M, asserted), 3 layers, dim 64, batch 1024,
    device-side edge dropout (the mask is read back and given to the oracle); on the column-sliced SpMM (the graph's
    9746 nodes fit the LDS) and on the edge-parallel gather SpMM bigger graphs take."""
    U, I, D, L, B = 6040, 3706, 64, 3, 1024
    adj = ml1m_like_graph()
    assert adj.nnz == 2 * 988_000 + U + I
    torch.manual_seed(3)
    eng = make_engine(U, I, D, L, "adam", 0.05, B, adj, dropout_rng="device", dropout_seed=11, spmm=spmm)
    assert (eng.model.graph()["slice_w"] == 4) == (spmm != "gather")
    w = get_weights(eng)
    rng = np.random.default_rng(1)
    batch = (rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B))
    loss, grads = eng.backward_only(tuple(torch.from_numpy(a) for a in batch))
    keep = eng.model.last_keep_mask().cpu().numpy().astype(bool)
    assert abs(keep.mean() - 0.6) < 0.005, keep.mean()
    dropped = olg.apply_edge_dropout(adj, keep, 0.6)
    loss_ref, g_ref = olg.lightgcn_grads(w, dropped, L, *batch, 1e-5)
    assert_scalar_close(loss, loss_ref, what="loss")
    with float64_oracle(olg):
        _, exact = olg.lightgcn_grads(to64(w), dropped.astype(np.float64), L, *batch, 1e-5)
    assert_grads_as_accurate(grads, g_ref, exact, "grad")
    eng2_mask_a = keep.copy()
    eng.backward_only(tuple(torch.from_numpy(a) for a in batch))
    keep_b = eng.model.last_keep_mask().cpu().numpy().astype(bool)
    assert (eng2_mask_a != keep_b).mean() > 0.3, "a fresh mask every step"
    scores = eng.model.predict(batch[0][:500], batch[1][:500]).cpu().numpy()
    assert_tensor_close(scores, olg.lightgcn_predict(w, adj, L, batch[0][:500], batch[1][:500]), 1e-5, "scores")
    with pytest.raises(IndexError):
        eng.train_single_batch((torch.tensor([0, U]), torch.tensor([0, 1]), torch.tensor([1, 2])))


@pytest.mark.parametrize("optimizer,lr", [("adam", 0.05), ("sgd", 0.1), ("rmsprop", 0.01)])
def test_optimizer_launch_stages_the_next_step(hip_device, optimizer, lr):
    """Sliced path, device draw: the optimizer launch of step t also draws step t + 1's edge streams and lays the fresh
    weights out sliced (hiprec_lightgcn_opt_stage).  Six consecutive steps against the oracle fed with the masks the
    engine used (read back statelessly after each step), an eval-mode predict and a load_state_dict in between (both
    invalidate what was staged), and the same run with `stage_next_step: False`: identical losses to 1e-6."""
    U, I, D, L, B = 300, 200, 64, 2, 256
    rng = np.random.default_rng(3)
    rows, cols = rng.integers(0, U, 4000), rng.integers(0, I, 4000)
    a = sp.coo_matrix((np.ones(4000, dtype=np.float32), (rows, cols + U)), shape=(U + I, U + I)).tocsr()
    a.data[:] = 1.0
    a = (a + a.T + sp.eye(U + I, dtype=np.float32, format="csr")).tocsr()
    a.data[:] = 1.0
    adj = sp.diags(1.0 / np.asarray(a.sum(1)).flatten()).dot(a).astype(np.float32).tocsr()
    adj.sort_indices()      # the keep bytes are in the engine's CSR order: rows, then ascending columns
    batches = [(rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)) for _ in range(6)]
    runs = {}
    for staged in (True, False):
        torch.manual_seed(3)
        eng = make_engine(U, I, D, L, optimizer, lr, B, adj, dropout_rng="device", dropout_seed=5, stage_next_step=staged)
        assert eng.model.graph()["slice_w"] == 4
        w = get_weights(eng)
        w0 = {k: v.copy() for k, v in w.items()}
        st = olg.new_opt_state(w, optimizer)
        from beta_recsys_amd import _lib

        lib, calls = _lib.load(), []
        plain = lib.hiprec_lightgcn_step_values
        lib.hiprec_lightgcn_step_values = lambda *a: (calls.append(eng.model._step), plain(*a))[1]   # steps that prepare themselves
        losses = []
        for s, batch in enumerate(batches):
            if s == 3:      # an evaluation between two training steps overwrites the staged buffers
                eng.model.predict(batch[0][:10], batch[1][:10])
                eng.model.train()
            if s == 4:      # new weights: what the optimizer launch laid out is stale
                load_weights(eng, w)
            if s == 5:      # ... and so it is after an in-place write to a parameter
                with torch.no_grad():
                    eng.model.user_embedding.weight.mul_(1.0)
            loss = eng.train_single_batch(tuple(torch.from_numpy(x) for x in batch))
            keep = eng.model.last_keep_mask().cpu().numpy().astype(bool)
            ref, g_ref = olg.lightgcn_grads(w, olg.apply_edge_dropout(adj, keep, 0.6), L, *batch, 1e-5)
            olg.opt_step(w, g_ref, st, optimizer, lr)
            assert_scalar_close(loss, ref, what=f"loss of step {s} (staged={staged})")
            losses.append(loss)
        lib.hiprec_lightgcn_step_values = plain
        # staged: steps 0 (nothing staged yet), 3 (after the predict), 4 (after load_state_dict) and 5 (after an in-place
        # write to a parameter) prepare themselves; steps 1 and 2 find their edge streams and E0 staged
        assert calls == ([1, 4, 5, 6] if staged else [1, 2, 3, 4, 5, 6]), calls
        runs[staged] = (losses, get_weights(eng))
    for a_, b_ in zip(runs[True][0], runs[False][0]):
        assert abs(a_ - b_) <= 1e-6 * abs(b_)
