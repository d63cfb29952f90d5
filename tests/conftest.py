"""pytest configuration: the `gpu` marker and import paths."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def hip_device():
    """cuda:0 on the GPU box.  GPU tests must never fall back silently: no GPU -> hard failure."""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but torch.cuda.is_available() is False")
    return torch.device("cuda:0")
