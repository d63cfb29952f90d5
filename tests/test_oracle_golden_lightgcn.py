"""Pin oracle/lightgcn_numpy.py against golden vectors captured from the real reference
(oracle/gen_golden.py --lightgcn), including the per-step edge-dropout masks.  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import assert_scalar_close, assert_tensor_close, load_golden
from oracle import lightgcn_numpy as olg

CASES = ["lightgcn_adam", "lightgcn_sgd_d64"]


def golden_adj(g):
    N = int(g["meta"][0] + g["meta"][1])
    a = sp.csr_matrix((g["adj_val"], (g["adj_row"], g["adj_col"])), shape=(N, N))
    a.sort_indices()
    return a


def golden_mask(g, s):
    return np.unpackbits(g["masks"][s])[: int(g["nnz"])].astype(bool)


def params(g, prefix):
    return {k: g[f"{prefix}/{k}"].astype(np.float32).copy() for k in olg.KEYS}


@pytest.mark.parametrize("case", CASES)
def test_lightgcn_oracle_matches_reference(case):
    g = load_golden(case)
    U, I, D, L, B, n_steps, _ = (int(x) for x in g["meta"])
    keep, decay = float(g["keep"]), float(g["decay"])
    adj = golden_adj(g)
    # the oracle's own adjacency builder reproduces the reference's D^-1 (A + I)
    mine = olg.build_norm_adj(U, I, g["edge_users"], g["edge_items"])
    assert (abs(mine - adj) > 1e-7).nnz == 0
    for s in range(n_steps):
        w = params(g, f"w{s}")
        dropped = olg.apply_edge_dropout(adj, golden_mask(g, s), keep)
        loss, grads = olg.lightgcn_grads(w, dropped, L, g["users"][s], g["pos"][s], g["neg"][s], decay)
        assert_scalar_close(loss, g["losses"][s], what=f"loss step {s}")
        for k in olg.KEYS:
            assert_tensor_close(grads[k], g[f"g{s + 1}/{k}"], 2e-5, f"grad {k} step {s}")
    wf = params(g, f"w{n_steps}")
    scores = olg.lightgcn_predict(wf, adj, L, g["probe_users"], g["probe_items"])
    assert_tensor_close(scores, g["probe_scores"], 1e-5, "probe scores")
