"""GPU: device-side negative sampling and loader builders (csrc/sampler.hip, beta-recsys_amd/data.py)
bit-exact against oracle/sampler_numpy.py, laid out like the reference's loaders
(tests/golden/sampler_loaders.npz), plus support / distinctness / uniformity at ML-1M size."""
import contextlib
import io
import types

import numpy as np
import pytest
import torch

from oracle import sampler_numpy as osn
from test_oracle_golden_sampler import check_support, frame, pooled_chi2

pytestmark = pytest.mark.gpu


def hdata():
    from beta_recsys_amd import data as hip_data

    return hip_data


def as_data(g, U, I):
    import pandas as pd

    train = pd.DataFrame({"col_user": g["train_users"], "col_item": g["train_items"],
                          "col_rating": g["train_ratings"]})
    return types.SimpleNamespace(train=train, n_users=U, n_items=I)


@pytest.mark.parametrize("k,seed", [(1, 0), (1, 2**61 + 5), (3, 7)])
def test_sampler_is_bit_exact_against_the_oracle(hip_device, k, seed):
    g, U, I, positives = frame()
    want = osn.sample_negatives(g["train_users"], g["train_items"], U, I, k, seed)
    got = hdata().sample_negatives(g["train_users"], g["train_items"], U, I, k, seed, hip_device)
    assert got.dtype == torch.int64 and got.shape == (len(g["train_users"]), k)
    assert np.array_equal(got.cpu().numpy(), want)


def test_sampler_bit_exact_on_a_larger_ragged_frame(hip_device):
    rng = np.random.default_rng(4)
    U, I, n = 400, 257, 6000
    users = rng.integers(0, U, n)
    users[users == 17] = 18                                 # user 17 has no row at all (deg 0 elsewhere is fine)
    items = rng.integers(0, I, n)                           # duplicates in the frame are legal
    want = osn.sample_negatives(users, items, U, I, 5, 123)
    got = hdata().sample_negatives(users, items, U, I, 5, 123, hip_device).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [1, 2, 3, 1000, 4099])
def test_device_permutation_matches_oracle_feistel(hip_device, n):
    from beta_recsys_amd import _lib

    lib = _lib.load()
    for seed in (0, 9, 2**61 + 12345):
        out = torch.empty(n, dtype=torch.int64, device=hip_device)
        _lib.check(lib.hiprec_random_permutation(_lib.ptr(out), n, seed, _lib.stream_ptr(hip_device)))
        assert np.array_equal(out.cpu().numpy(), osn.random_permutation(n, seed))


def test_loader_builders_have_the_reference_layout(hip_device):
    g, U, I, positives = frame()
    data = as_data(g, U, I)
    hd = hdata()
    with contextlib.redirect_stdout(io.StringIO()) as log:
        bpr = hd.instance_bpr_loader(data, 16, "cuda:0", seed=1)
        bce = hd.instance_bce_loader(data, 16, "cuda:0", 3, seed=2)
        mul = hd.instance_mul_neg_loader(data, 16, "cuda:0", 3, seed=3)
    assert "Making PairwiseNegativeDataset of length 440" in log.getvalue()   # base_data.py:252
    assert "Making RatingDataset of length 1760" in log.getvalue()            # base_data.py:215
    # bpr: PairwiseNegativeDataset-like tensors in frame order (golden: the reference's own tensors)
    assert np.array_equal(bpr.user_tensor.cpu().numpy(), g["bpr_users"])
    assert np.array_equal(bpr.pos_item_tensor.cpu().numpy(), g["bpr_pos"])
    check_support(g["train_users"], bpr.neg_item_tensor.cpu().numpy(), positives, I)
    assert len(bpr) == len(range(0, 440, 16))
    # bce: row, then its negatives with rating 0
    bu, bi, br = (t.cpu().numpy() for t in bce.tensors)
    assert np.array_equal(bu, g["bce_users"])
    assert np.array_equal(bi.reshape(-1, 4)[:, 0], g["bce_items"].reshape(-1, 4)[:, 0])
    assert np.array_equal(br.reshape(-1, 4)[:, 0], g["bce_ratings"].reshape(-1, 4)[:, 0]) and not br.reshape(-1, 4)[:, 1:].any()
    check_support(g["train_users"], bi.reshape(-1, 4)[:, 1:], positives, I)
    batches = list(bce)
    assert len(batches) == 110 and all(len(b) == 3 for b in batches)
    seen = torch.cat([b[0] * I + b[1] for b in batches]).cpu().numpy()
    assert sorted(seen.tolist()) == sorted((bu * I + bi).tolist())          # an epoch visits every sample once
    assert batches[0][2].dtype == torch.float32
    # mul_neg: [n, k] negatives
    mu, mp, mn = (t.cpu().numpy() for t in mul.tensors)
    assert mn.shape == g["mul_neg"].shape and np.array_equal(mu, g["mul_users"]) and np.array_equal(mp, g["mul_pos"])
    check_support(g["train_users"], mn, positives, I)
    assert next(iter(mul))[2].shape == (16, 3)
    # user 5 has 3 untouched items: asking for 4 fails like random.sample does (base_data.py:198)
    with pytest.raises(ValueError, match="Sample larger than population"):
        hd.instance_bce_loader(data, 16, "cuda:0", 4)
    bad = as_data(g, U, I - 1)                                              # an item id outside the pool
    with pytest.raises(IndexError):
        hd.instance_bpr_loader(bad, 16, "cuda:0")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hd.instance_bpr_loader(data, 16, "cpu")


def test_device_draws_are_uniform_like_the_reference(hip_device):
    g, U, I, positives = frame()
    users = g["train_users"]
    ours = np.stack([hdata().sample_negatives(users, g["train_items"], U, I, 1, seed, hip_device)[:, 0].cpu().numpy()
                     for seed in range(100, 160)])
    z = pooled_chi2(users, ours, positives, I)
    assert max(abs(v) for v in z.values()) < 5.0
    z_ref = pooled_chi2(users, g["bpr_neg_draws"].astype(np.int64), positives, I)
    assert max(abs(v) for v in z_ref.values()) < 5.0


def test_bpr_loader_feeds_the_engine(hip_device):
    import beta_recsys_amd as hp

    g, U, I, positives = frame()
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=16, device_str="cuda:0", optimizer="adam", lr=0.01,
                         batch_size=64, loss="bpr"), "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = hp.MFEngine(cfg)
        loader = hdata().instance_bpr_loader(as_data(g, U, I), 64, "cuda:0")
        losses = []
        for epoch in range(3):
            eng.train_an_epoch(loader, epoch)
            losses.append(eng.epoch_stats().loss_sum)
    assert all(np.isfinite(losses)) and losses[2] < losses[0]


def test_full_size_support_distinctness_uniformity(hip_device):
    """ML-1M-shaped frame: 1 M rows over 6040 x 3706, 4 negatives per row."""
    gen = torch.Generator().manual_seed(3)
    U, I, n, k = 6040, 3706, 1_000_000, 4
    users = torch.randint(0, U, (n,), generator=gen)
    p = 1.0 / torch.arange(1, I + 1, dtype=torch.float64)
    items = torch.multinomial(p / p.sum(), n, True, generator=gen)
    neg = hdata().sample_negatives(users.to(hip_device), items.to(hip_device), U, I, k, 11)
    assert neg.shape == (n, k) and int(neg.min()) >= 0 and int(neg.max()) < I
    pos_keys = torch.unique(users.to(hip_device) * I + items.to(hip_device))
    neg_keys = users.to(hip_device)[:, None] * I + neg
    assert not bool(torch.isin(neg_keys.reshape(-1), pos_keys).any())       # never a positive of the user
    srt = neg.sort(dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())                          # distinct within a row
    # uniformity over items a user can still draw: popular items are positives of many users, so
    # compare per-item draw counts with the number of draws for which the item was eligible
    counts = torch.bincount(neg.reshape(-1), minlength=I).double().cpu()
    deg = torch.bincount(torch.div(pos_keys, I, rounding_mode="floor"), minlength=U).cpu()
    rows_per_user = torch.bincount(users, minlength=U).double()
    # expected draws of item i = sum over rows of k / (I - deg(u)) for rows whose user does not own i
    w = rows_per_user * k / (I - deg).double()
    owned = torch.zeros(U, I, dtype=torch.bool)
    owned[torch.div(pos_keys, I, rounding_mode="floor").cpu(), (pos_keys % I).cpu()] = True
    expected = (w[:, None] * (~owned)).sum(0)
    assert float(counts[expected == 0].sum()) == 0.0                        # owned by every active user
    live = expected > 50
    zscore = (counts[live] - expected[live]) / expected[live].sqrt()
    assert int(live.sum()) > 3000 and float(zscore.abs().max()) < 6.0 and abs(float(zscore.mean())) < 0.2
    again = hdata().sample_negatives(users.to(hip_device), items.to(hip_device), U, I, k, 11)
    assert torch.equal(again, neg)


def test_bce_loader_runs_the_resident_bce_epoch(hip_device):
    """MF with loss 'bce' fed by the device-side instance_bce_loader (1 positive : 4 negatives per
    row, data/base_data.py:182-216): the engine stages the (user, item, rating) stream and runs the
    resident epoch; the loss falls and the step counter equals the number of batches."""
    import beta_recsys_amd as hp

    g, U, I, positives = frame()
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=16, device_str="cuda:0", optimizer="adam", lr=0.01,
                         batch_size=128, loss="bce"), "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = hp.MFEngine(cfg)
        loader = hdata().instance_bce_loader(as_data(g, U, I), 128, "cuda:0", 2)
        assert eng.prepare_epoch(loader) is not None          # resident path, not the per-batch loop
        losses = []
        for epoch in range(4):
            eng.train_an_epoch(loader, epoch)
            losses.append(eng.epoch_stats().loss_sum)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert eng.epoch_stats().step == 4 * len(loader)
