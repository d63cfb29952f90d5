"""GPU parity tests of LightGCN: CSR SpMM, propagation, loss/backward and the full step vs golden
vectors from the real reference (same dropped edges for the same torch seed) and vs the
numpy/scipy oracle at BASELINE's C5 shape (ML-1M-sized graph, 3 layers, dim 64, batch 1024)."""
import contextlib
import ctypes
import io

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from helpers import EPS32, assert_scalar_close, assert_tensor_close, load_golden
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


@pytest.mark.parametrize("case", ["lightgcn_adam", "lightgcn_sgd_d64"])
def test_lightgcn_step_matches_reference(hip_device, case):
    g = load_golden(case)
    U, I, D, L, B, n_steps, seed = (int(x) for x in g["meta"])
    opt, lr, keep, decay = str(g["optimizer"]), float(g["lr"]), float(g["keep"]), float(g["decay"])
    adj = golden_adj(g)
    torch.manual_seed(seed)
    eng = make_engine(U, I, D, L, opt, lr, B, adj, keep, decay)
    w_init = get_weights(eng)
    for k in w_init:  # same seed -> the reference's xavier init, bit for bit
        assert np.array_equal(w_init[k], g[f"w0/{k}"]), k
    assert list(w_init.keys()) == ["user_embedding.weight", "item_embedding.weight"]
    for s in range(n_steps):
        batch = tuple(torch.from_numpy(g[k][s]) for k in ("users", "pos", "neg"))
        load_weights(eng, params(g, f"w{s}"))
        torch.manual_seed(1000 + s)
        loss, grads = eng.backward_only(batch)
        kept = eng.model._ws["keep"][: int(g["nnz"])].cpu().numpy().astype(bool)
        assert np.array_equal(kept, golden_mask(g, s)), "same seed must drop the same edges"
        assert_scalar_close(loss, g["losses"][s], what=f"loss step {s}")
        for k in olg.KEYS:
            assert_tensor_close(grads[k].cpu().numpy(), g[f"g{s + 1}/{k}"], 2e-5, f"grad {k} step {s}")
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
            assert_scalar_close(loss, g["losses"][s], 5e-5, f"trajectory loss {s}")
        w = get_weights(eng)
        for k in olg.KEYS:
            frac_bad = np.mean(np.abs(w[k] - g[f"w{n_steps}/{k}"]) > 1e-3 * lr + 1e-6)
            assert frac_bad < 0.02, f"{k}: {frac_bad:.2%} off trajectory"
    load_weights(eng, params(g, f"w{n_steps}"))
    scores = eng.model.predict(g["probe_users"], g["probe_items"])
    assert_tensor_close(scores.cpu().numpy(), g["probe_scores"], 1e-5, "probe scores")
    assert not eng.model.training
    ue, ie = eng.model.forward()  # eval mode: no dropout
    ref_u, ref_i = olg.propagate(params(g, f"w{n_steps}"), adj, L)
    assert_tensor_close(ue.cpu().numpy(), ref_u, 1e-5, "propagated users")
    assert_tensor_close(ie.cpu().numpy(), ref_i, 1e-5, "propagated items")


def ml1m_like_graph(seed=0, U=6040, I=3706, n_edges=1_000_000):
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, I + 1) ** 0.9
    p /= p.sum()
    eu = rng.integers(0, U, n_edges)
    ei = rng.permutation(I)[rng.choice(I, n_edges, p=p)]
    return olg.build_norm_adj(U, I, eu, ei)


def test_lightgcn_full_size_c5_vs_oracle(hip_device):
    """BASELINE configs[4] shape: ~1M interactions (nnz ~2M), 3 layers, dim 64, batch 1024,
    device-side edge dropout (the mask is read back and given to the oracle)."""
    U, I, D, L, B = 6040, 3706, 64, 3, 1024
    adj = ml1m_like_graph()
    torch.manual_seed(3)
    eng = make_engine(U, I, D, L, "adam", 0.05, B, adj, dropout_rng="device", dropout_seed=11)
    w = get_weights(eng)
    rng = np.random.default_rng(1)
    batch = (rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B))
    loss, grads = eng.backward_only(tuple(torch.from_numpy(a) for a in batch))
    keep = eng.model._ws["keep"][: adj.nnz].cpu().numpy().astype(bool)
    assert abs(keep.mean() - 0.6) < 0.005, keep.mean()
    dropped = olg.apply_edge_dropout(adj, keep, 0.6)
    loss_ref, g_ref = olg.lightgcn_grads(w, dropped, L, *batch, 1e-5)
    assert_scalar_close(loss, loss_ref, what="loss")
    for k in olg.KEYS:
        assert_tensor_close(grads[k].cpu().numpy(), g_ref[k], 2e-5, f"grad {k}")
    eng2_mask_a = keep.copy()
    eng.backward_only(tuple(torch.from_numpy(a) for a in batch))
    keep_b = eng.model._ws["keep"][: adj.nnz].cpu().numpy().astype(bool)
    assert (eng2_mask_a != keep_b).mean() > 0.3, "a fresh mask every step"
    scores = eng.model.predict(batch[0][:500], batch[1][:500]).cpu().numpy()
    assert_tensor_close(scores, olg.lightgcn_predict(w, adj, L, batch[0][:500], batch[1][:500]), 1e-5, "scores")
    with pytest.raises(IndexError):
        eng.train_single_batch((torch.tensor([0, U]), torch.tensor([0, 1]), torch.tensor([1, 2])))
