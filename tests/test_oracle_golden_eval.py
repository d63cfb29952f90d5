"""CPU: pin the numpy restatement of the ranking evaluation (oracle/eval_numpy.py) against the
outputs of the REAL reference's evaluate() (tests/golden/eval_*.npz, oracle/gen_golden.py --eval)."""
import numpy as np
import pytest

from helpers import load_golden
from oracle import eval_numpy as oev

EVAL_FIXTURES = ("eval_leave_one_out", "eval_ragged", "eval_ties", "eval_no_hits", "eval_full_catalogue")
TOL = 1e-12   # fp64 sums of a few dozen terms; only the summation order over users differs


@pytest.mark.parametrize("name", EVAL_FIXTURES)
def test_oracle_matches_reference_evaluate(name):
    g = load_golden(name)
    n_common, table = oev.rank_metrics(g["users"], g["ratings"], g["scores"], g["k_list"].tolist())
    np.testing.assert_allclose(table, g["metrics"], rtol=TOL, atol=TOL)
    has_pos = {u for u, r in zip(g["users"], g["ratings"]) if r >= 1}
    assert n_common == len(has_pos)


def test_oracle_evaluate_keys_follow_reference_layout():
    g = load_golden("eval_leave_one_out")
    res = oev.evaluate(g["users"], g["ratings"], g["scores"], ["ndcg", "recall"], [5, 10])
    assert list(res) == ["ndcg@5", "recall@5", "ndcg@10", "recall@10"]
    ks = g["k_list"].tolist()
    assert res["ndcg@10"] == pytest.approx(g["metrics"][ks.index(10), 2], rel=TOL)
    single = oev.evaluate(g["users"], g["ratings"], g["scores"], ["map"], 5)   # int k (eval_engine.py:81-82)
    assert list(single) == ["map@5"]


def test_oracle_tie_order_is_first_occurrence():
    # two candidates with the same score: the one that appears first in the frame ranks first
    users = [0, 0, 0]
    scores = [0.5, 0.5, 0.1]
    _, first = oev.rank_metrics(users, [1, 0, 0], scores, [1])
    _, second = oev.rank_metrics(users, [0, 1, 0], scores, [1])
    assert first[0, 0] == 1.0 and second[0, 0] == 0.0
