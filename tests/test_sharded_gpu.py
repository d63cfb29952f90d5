"""The row-sharded engine with the REAL kernel backend (libhiprec) on one MI355X: a world-size-1
process group over RCCL runs every exchange (all-to-all with itself) and every kernel of the
sharded step; the result must equal the oracle's single-process step."""
import contextlib
import io
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from helpers import KEYS, assert_scalar_close
from oracle import mf_numpy as onp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group(hip_device):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=hip_device)
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("optimizer,lr", [("sgd", 0.1), ("adam", 0.05)])
def test_sharded_step_with_hip_kernels(nccl_group, optimizer, lr):
    from beta_recsys_amd.sharded import ShardedMFEngine

    U, I, D, B = 300, 200, 64, 512
    w0 = onp.init_params(U, I, D, seed=3)
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, device_str="cuda:0", optimizer=optimizer,
                         lr=lr, batch_size=B, loss="bpr"),
           "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    with contextlib.redirect_stdout(io.StringIO()):
        eng = ShardedMFEngine(cfg, full_state={k: torch.from_numpy(v) for k, v in w0.items()})
    w = onp.copy_params(w0)
    st = onp.new_opt_state(w, optimizer)
    rng = np.random.default_rng(0)
    for _ in range(3):
        batch = (rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B))
        batch[1][: B // 4] = batch[1][0]
        loss, reg = eng.train_single_batch(tuple(torch.from_numpy(a) for a in batch))
        ref_loss, ref_reg = onp.mf_train_step(w, st, batch, "bpr", optimizer, lr)
        assert_scalar_close(loss, ref_loss, 2e-5, "loss")
        assert_scalar_close(reg, ref_reg, 2e-5, "reg")
    full = eng.gather_full_state_dict()
    tol = 1e-6 if optimizer == "sgd" else 2e-3
    for k in KEYS:
        frac_bad = np.mean(np.abs(full[k].cpu().numpy() - w[k]) > tol)
        assert frac_bad < 0.01, f"{k}: {frac_bad:.2%} differ"
