"""Pin the PyTorch-CPU ports of the sibling models (oracle/torch_port.py: the cpu_baseline leg of
bench.py --workload pgmf|t2v|ngcf) on the goldens captured from the real reference.  CPU only."""
import numpy as np
import pytest
import torch

from helpers import assert_scalar_close, assert_tensor_close, load_golden
from oracle.torch_port import TorchNGCFPort, TorchPGMFPort, TorchT2VPort
from test_oracle_golden_ngcf import ngcf_adj, ngcf_batch, ngcf_params
from test_oracle_golden_pgmf import pgmf_params
from test_oracle_golden_t2v import t2v_batch, t2v_params


@pytest.mark.parametrize("case", ["pgmf_adam", "pgmf_sgd_clip"])
def test_pgmf_port_follows_the_reference_trajectory(case):
    g = load_golden(case)
    n_steps = int(g["meta"][4])
    port = TorchPGMFPort(pgmf_params(g, "w0"), str(g["optimizer"]), float(g["lr"]), float(g["l2_lambda"]),
                         float(g["grad_clip"]))
    for s in range(n_steps):
        assert_scalar_close(port.step((g["users"][s], g["pos"][s], g["neg"][s])), g["losses"][s], 2e-6, f"loss {s}")
    w = port.numpy_weights()
    for k in w:
        assert_tensor_close(w[k], g[f"w{n_steps}/{k}"], 1e-5, f"final {k}")


@pytest.mark.parametrize("case", ["t2v_adam", "t2v_sgd_d100"])
def test_t2v_port_follows_the_reference_trajectory(case):
    g = load_golden(case)
    B, n_steps = int(g["meta"][3]), int(g["meta"][5])
    port = TorchT2VPort(t2v_params(g, "w0"), B, str(g["optimizer"]), float(g["lr"]))
    for s in range(n_steps):
        assert_scalar_close(port.step(t2v_batch(g, s)), g["losses"][s], 2e-6, f"loss {s}")
    w = port.numpy_weights()
    for k in w:
        assert_tensor_close(w[k], g[f"w{n_steps}/{k}"], 1e-5, f"final {k}")


@pytest.mark.parametrize("case", ["ngcf_adam", "ngcf_sgd_widths"])
def test_ngcf_port_follows_the_reference_trajectory(case):
    g = load_golden(case)
    B, n_steps = int(g["meta"][4]), int(g["meta"][5])
    co = ngcf_adj(g).tocoo()
    adj = torch.sparse_coo_tensor(torch.from_numpy(np.vstack((co.row, co.col)).astype(np.int64)),
                                  torch.from_numpy(co.data), torch.Size(co.shape))
    port = TorchNGCFPort(ngcf_params(g, "w0"), adj, [float(p) for p in g["mess_dropout"]], float(g["decay"]), B,
                         str(g["optimizer"]), float(g["lr"]))
    for s in range(n_steps):
        torch.manual_seed(2000 + s)
        assert_scalar_close(port.step(ngcf_batch(g, s)), g["losses"][s], 5e-6, f"loss {s}")
    w = port.numpy_weights()
    for k in w:
        assert_tensor_close(w[k], g[f"w{n_steps}/{k}"], 2e-5, f"final {k}")


@pytest.mark.parametrize("case", ["ncf_neumf_adam"])
def test_neumf_port_follows_the_reference_trajectory(case):
    """oracle/torch_port.py::TorchNeuMFPort (the cpu_baseline of bench.py --workload ncf) on the golden captured from
    the real NeuMFEngine: per-step losses and the final weights."""
    from oracle.torch_port import TorchNeuMFPort
    from test_oracle_golden_ncf import params

    g = load_golden(case)
    n_steps = int(g["meta"][5])
    port = TorchNeuMFPort(params(g, "w0"), str(g["optimizer"]), float(g["lr"]))
    for s in range(n_steps):
        assert_scalar_close(port.step((g["users"][s], g["items"][s], g["ratings"][s])), g["losses"][s], 2e-6, f"loss {s}")
    w = port.numpy_weights()
    for k in w:
        assert_tensor_close(w[k], g[f"w{n_steps}/{k}"], 1e-5, f"final {k}")


@pytest.mark.parametrize("case", ["lightgcn_adam", "lightgcn_sgd_d64"])
def test_lightgcn_port_follows_the_reference_trajectory(case):
    """oracle/torch_port.py::TorchLightGCNPort (the cpu_baseline of bench.py --workload lightgcn): with the reference's
    torch seeds it drops the same edges (asserted against the golden's masks) and reproduces losses and weights."""
    from oracle.torch_port import TorchLightGCNPort
    from test_oracle_golden_lightgcn import golden_adj, golden_mask, params

    g = load_golden(case)
    U, I, D, L, B, n_steps, seed = (int(x) for x in g["meta"])
    co = golden_adj(g).tocoo()
    adj = torch.sparse_coo_tensor(torch.from_numpy(np.vstack((co.row, co.col)).astype(np.int64)),
                                  torch.from_numpy(co.data.astype(np.float32)), torch.Size(co.shape))
    port = TorchLightGCNPort(params(g, "w0"), adj, L, float(g["keep"]), float(g["decay"]), str(g["optimizer"]),
                             float(g["lr"]))
    for s in range(n_steps):
        torch.manual_seed(1000 + s)
        drawn = (torch.rand(int(g["nnz"])) + float(g["keep"])).int().bool().numpy()
        assert np.array_equal(drawn, golden_mask(g, s)), "the port's draw is the reference's"
        torch.manual_seed(1000 + s)
        assert_scalar_close(port.step(tuple(g[k][s] for k in ("users", "pos", "neg"))), g["losses"][s], 5e-6, f"loss {s}")
    w = port.numpy_weights()
    for k in w:
        assert_tensor_close(w[k], g[f"w{n_steps}/{k}"], 2e-5, f"final {k}")
