"""does a node reordering speed up the LightGCN SpMM? (same graph, permuted ids; kernel unchanged)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from beta_recsys_amd import _lib
from beta_recsys_amd.lightgcn import _csr_from_coo, _slice_rows

U, I, D = 6040, 3706, 64
rng = np.random.default_rng(0)
n_edges = 1_000_000
p = 1.0 / np.arange(1, I + 1) ** 0.9
eu = rng.integers(0, U, n_edges)
ei = rng.permutation(I)[rng.choice(I, n_edges, p=p / p.sum())]
N = U + I
lib = _lib.load()
dev = torch.device("cuda:0")


def time_order(name, perm):
    """perm[old] = new id"""
    rows, cols = np.concatenate([eu, ei + U]), np.concatenate([ei + U, eu])
    rows, cols = perm[rows], perm[cols]
    a = sp.coo_matrix((np.ones(len(rows), dtype=np.float32), (rows, cols)), shape=(N, N)).tocsr()
    a.data[:] = 1.0
    a = (a + sp.eye(N, dtype=np.float32, format="csr")).tocoo()
    r, c, v = (torch.from_numpy(x) for x in (a.row.astype(np.int64), a.col.astype(np.int64), a.data))
    rp, cc, vv, _ = _csr_from_coo(r, c, v, N, dev)
    sl = _slice_rows(rp, cc, vv, N, a.nnz)
    csr = _lib.Csr(rp.data_ptr(), cc.data_ptr(), vv.data_ptr(), None, N, a.nnz, sl.data_ptr())
    x = torch.randn(N, D, device=dev)
    y = torch.zeros(N, D, device=dev)
    acc = torch.zeros(N, D, device=dev)
    st = _lib.stream_ptr(dev)
    args = (ctypes.byref(csr), None, 1.0, _lib.ptr(x), _lib.ptr(y), _lib.ptr(acc), D, st)
    for _ in range(5):
        _lib.check(lib.hiprec_spmm_csr(*args))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        _lib.check(lib.hiprec_spmm_csr(*args))
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} nnz {a.nnz}  {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us / SpMM", flush=True)


ident = np.arange(N)
time_order("natural order", ident)
deg_u, deg_i = np.bincount(eu, minlength=U), np.bincount(ei, minlength=I)
perm = ident.copy()
perm[np.argsort(-deg_i) + U] = np.arange(I) + U
time_order("items by popularity", perm)
perm2 = perm.copy()
perm2[np.argsort(-deg_u)] = np.arange(U)
time_order("items + users by degree", perm2)
# users clustered by their most popular item (cheap community proxy)
rank_i = np.empty(I, dtype=np.int64); rank_i[np.argsort(-deg_i)] = np.arange(I)
best = np.full(U, I, dtype=np.int64)
np.minimum.at(best, eu, rank_i[ei])
second = np.full(U, I, dtype=np.int64)
perm3 = perm.copy()
perm3[np.lexsort((deg_u, best))] = np.arange(U)
time_order("users clustered by top item", perm3)
rnd = ident.copy(); rnd[:U] = rng.permutation(U); rnd[U:] = rng.permutation(I) + U
time_order("random order", rnd)
