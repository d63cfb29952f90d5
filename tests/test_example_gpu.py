"""GPU: the end-to-end example (device-side negative sampling -> fused epochs -> on-device validation)
runs and learns on a small problem."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_end_to_end_example_learns(hip_device):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    try:
        import mf_end_to_end
    finally:
        sys.path.pop(0)
    hist = mf_end_to_end.main(["--users", "400", "--items", "300", "--interactions", "20000", "--emb-dim", "32",
                               "--batch-size", "512", "--epochs", "6", "--lr", "0.02", "--eval-negatives", "50"])
    assert len(hist) == 6
    assert hist[-1]["loss"] < hist[0]["loss"]
    # the synthetic data only carries item popularity: any trained model must rank the held-out
    # positive far above the ~0.11 ndcg@10 of a random order over 51 candidates
    assert min(h["ndcg@10"] for h in hist) > 0.25
    assert 0.0 <= hist[-1]["recall@20"] <= 1.0


def test_sibling_models_example_learns(hip_device):
    """NGCF, Triple2vec and PairwiseGMF on data with planted user / item groups: each one's loss falls and
    its ranking of the held-out positive (placed LAST in its block, so ties count against it) is far above
    the ~0.09 ndcg@10 of a random order over 51 candidates (measured: 0.66 / 0.67 / 0.67)."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    try:
        import siblings_end_to_end
    finally:
        sys.path.pop(0)
    hist = siblings_end_to_end.main(["--users", "600", "--items", "400", "--interactions", "30000", "--emb-dim", "16",
                                     "--epochs", "5", "--eval-negatives", "50"])
    assert set(hist) == {"ngcf", "triple2vec", "pairwise_gmf"}
    for name, h in hist.items():
        assert len(h) == 5
        assert h[-1]["loss"] < h[0]["loss"], name
        assert max(x["ndcg@10"] for x in h) > 0.4, (name, [x["ndcg@10"] for x in h])
