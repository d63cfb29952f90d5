"""CPU: the negative-sampling oracle (oracle/sampler_numpy.py) against what the REAL reference's loader
builders produce (tests/golden/sampler_loaders.npz, oracle/gen_golden.py --sampler): identical row
layout, identical support (never a positive of the user, inside the pool, distinct within a row),
and the same — uniform — distribution over each user's untouched items."""
import numpy as np
import pytest

from helpers import load_golden
from oracle import sampler_numpy as osn


def frame():
    g = load_golden("sampler_loaders")
    U, I = (int(v) for v in g["meta"])
    positives = [set() for _ in range(U)]
    for u, i in zip(g["train_users"], g["train_items"]):
        positives[int(u)].add(int(i))
    return g, U, I, positives


def check_support(users, negs, positives, n_items):
    negs = np.asarray(negs).reshape(len(users), -1)
    for u, row in zip(users, negs):
        assert len(set(row.tolist())) == len(row), "negatives of one row must be distinct (random.sample)"
        assert not (set(row.tolist()) & positives[int(u)]), "a negative is one of the user's positives"
        assert row.min() >= 0 and row.max() < n_items


def test_reference_loader_layout_and_support():
    g, U, I, positives = frame()
    # instance_bpr_loader: rows in frame order, one negative each (base_data.py:238-246)
    assert np.array_equal(g["bpr_users"], g["train_users"]) and np.array_equal(g["bpr_pos"], g["train_items"])
    check_support(g["bpr_users"], g["bpr_neg"], positives, I)
    # instance_bce_loader: row, then its 3 negatives with rating 0 (base_data.py:201-209)
    k = 3
    assert np.array_equal(g["bce_users"], np.repeat(g["train_users"], k + 1))
    items = g["bce_items"].reshape(-1, k + 1)
    ratings = g["bce_ratings"].reshape(-1, k + 1)
    assert np.array_equal(items[:, 0], g["train_items"]) and np.array_equal(ratings[:, 0], g["train_ratings"])
    assert not ratings[:, 1:].any()
    check_support(g["train_users"], items[:, 1:], positives, I)
    # instance_mul_neg_loader: neg tensor [n, k]
    assert g["mul_neg"].shape == (len(g["train_users"]), k)
    check_support(g["mul_users"], g["mul_neg"], positives, I)


@pytest.mark.parametrize("k,seed", [(1, 0), (1, 99), (3, 7)])
def test_oracle_has_the_reference_support(k, seed):
    g, U, I, positives = frame()
    neg = osn.sample_negatives(g["train_users"], g["train_items"], U, I, k, seed)
    assert neg.shape == (len(g["train_users"]), k)
    check_support(g["train_users"], neg, positives, I)
    again = osn.sample_negatives(g["train_users"], g["train_items"], U, I, k, seed)
    assert np.array_equal(neg, again)                                   # pure function of the seed
    other = osn.sample_negatives(g["train_users"], g["train_items"], U, I, k, seed + 1)
    assert (neg != other).mean() > 0.5


def pooled_chi2(users, draws, positives, n_items):
    """Per user: chi-square of the pooled draws against uniform over the user's untouched items,
    returned as (statistic - dof) / sqrt(2 dof) (approximately standard normal under H0)."""
    z = {}
    for u in sorted(set(users.tolist())):
        pool = sorted(set(range(n_items)) - positives[u])
        if len(pool) < 2:
            continue
        col = draws[:, users == u].reshape(-1)
        counts = np.array([(col == it).sum() for it in pool], dtype=np.float64)
        assert counts.sum() == col.size                       # nothing outside the pool
        expected = col.size / len(pool)
        stat = ((counts - expected) ** 2 / expected).sum()
        dof = len(pool) - 1
        z[u] = (stat - dof) / np.sqrt(2 * dof)
    return z


def test_oracle_and_reference_draw_from_the_same_uniform_distribution():
    g, U, I, positives = frame()
    users = g["train_users"]
    ref_draws = g["bpr_neg_draws"].astype(np.int64)                     # 400 independent reference loaders
    ours = np.stack([osn.sample_negatives(users, g["train_items"], U, I, 1, seed)[:, 0] for seed in range(60)])
    z_ref = pooled_chi2(users, ref_draws, positives, I)
    z_our = pooled_chi2(users, ours, positives, I)
    assert max(abs(v) for v in z_ref.values()) < 5.0                    # the reference is uniform ...
    assert max(abs(v) for v in z_our.values()) < 5.0                    # ... and so is the restated generator
    # user 5 has exactly 3 untouched items: all of them get drawn, by both
    pool5 = set(range(I)) - positives[5]
    assert len(pool5) == 3
    assert set(ref_draws[:, users == 5].reshape(-1).tolist()) == pool5
    assert set(ours[:, users == 5].reshape(-1).tolist()) == pool5


def test_oracle_raises_like_random_sample_when_a_user_runs_out():
    g, U, I, positives = frame()
    with pytest.raises(ValueError):
        osn.sample_negatives(g["train_users"], g["train_items"], U, I, 4, 0)   # user 5 has 3 untouched items


def test_rank_to_item_map_and_feistel_bijection():
    rng = np.random.default_rng(1)
    for _ in range(50):
        n_items = int(rng.integers(2, 60))
        pos = np.sort(rng.permutation(n_items)[: int(rng.integers(0, n_items))])
        missing = [i for i in range(n_items) if i not in set(pos.tolist())]
        for r, want in enumerate(missing):
            assert osn.missing_item(pos, r, n_items) == want
            # the kernel's search: smallest t with pos[t] - t > r
            t = next((t for t in range(len(pos)) if pos[t] - t > r), len(pos))
            assert r + t == want
    for n in (1, 2, 3, 17, 64, 65, 1000, 4099):
        for seed in (0, 1, 2**61 + 12345):
            assert sorted(osn.random_permutation(n, seed).tolist()) == list(range(n))
