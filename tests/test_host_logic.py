"""CPU-only tests: the C-ABI library loads and exports every symbol include/hiprec.h declares, and
the host-side mirror of the reference interface behaves like beta_rec.models.{torch_engine,mf}
where no compute is involved.  No kernel is launched here."""
import contextlib
import ctypes
import io
import os
import re

import numpy as np
import pytest
import torch

from helpers import KEYS, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_engine(U=11, I=7, D=5, optimizer="sgd", loss="bpr", **extra):
    import beta_recsys_amd as hp

    model = dict(n_users=U, n_items=I, emb_dim=D, device_str="cpu", optimizer=optimizer, lr=0.05,
                 batch_size=4, loss=loss)
    cfg = {"model": model, "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    cfg.update(extra)
    with contextlib.redirect_stdout(io.StringIO()):
        return hp.MFEngine(cfg)


def declared_functions():
    text = open(os.path.join(ROOT, "include", "hiprec.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hiprec_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from beta_recsys_amd import _lib

    names = declared_functions()
    assert len(names) >= 15
    lib = _lib.load()
    for n in names:
        assert hasattr(lib, n), f"libhiprec.so does not export {n}"
        assert n in _lib.SIGNATURES, f"_lib.py has no prototype for {n}"
    assert set(_lib.SIGNATURES) == set(names), "prototypes for symbols the header does not declare"
    assert lib.hiprec_version() == 100
    assert lib.hiprec_stats_bytes() == ctypes.sizeof(_lib.Stats)
    assert lib.hiprec_scratch_bytes(4096) >= 16


def test_bad_arguments_return_error_codes_without_a_gpu():
    """Argument validation happens before any HIP call, so it can be checked on the CPU box."""
    from beta_recsys_amd import _lib

    lib = _lib.load()
    rc = lib.hiprec_gather_rows(None, 10, 4, None, 5, None, None, None)
    assert rc == -1 and b"NULL" in lib.hiprec_last_error()
    rc = lib.hiprec_opt_dense_step(7, None, None, None, None, 4, 0.1, 0.9, 0.999, 1e-8, None, None, -1,
                                   None)
    assert rc == -1
    with pytest.raises(_lib.HiprecError):
        _lib.check(rc)
    t = _lib.MfTables(0, 0, 0, 0, 0, 4, 5, 8, 0)
    rc = lib.hiprec_mf_predict(ctypes.byref(t), None, None, 3, None, None, None)
    assert rc == -1 and b"NULL tensor pointer" in lib.hiprec_last_error()


def test_newer_entry_points_validate_before_touching_the_gpu():
    """Same contract for the eval / sampler / fused-step / shard / dp entry points."""
    from beta_recsys_amd import _lib

    lib = _lib.load()
    ks = (ctypes.c_int32 * 2)(5, 10)
    rc = lib.hiprec_rank_metrics(None, 4, None, None, ks, 0, None, 0, None, None)
    assert rc == -1 and b"n_k=0" in lib.hiprec_last_error()
    rc = lib.hiprec_rank_metrics(None, 4, None, None, ks, 9, None, 0, None, None)
    assert rc == -1 and b"outside" in lib.hiprec_last_error()
    bad_k = (ctypes.c_int32 * 2)(5, 0)
    out = (ctypes.c_double * 8)()
    rc = lib.hiprec_rank_metrics(None, 4, None, None, bad_k, 2, None, 0, out, None)
    assert rc == -1 and b"k[1]=0" in lib.hiprec_last_error()
    assert lib.hiprec_rank_metrics_workspace_bytes(100, 2) >= 100 * 2 * 8

    rc = lib.hiprec_sample_negatives(None, None, 0, 10, None, 5, 1, 1, None, None, None)
    assert rc == -1 and b"bad sizes" in lib.hiprec_last_error()
    rc = lib.hiprec_sample_negatives(None, None, 3, 10, None, 5, 0, 1, None, None, None)
    assert rc == -1 and b"bad sizes" in lib.hiprec_last_error()
    rc = lib.hiprec_sample_negatives(None, None, 3, 10, None, 5, 1, 1, None, None, None)
    assert rc == -1 and b"NULL" in lib.hiprec_last_error()

    rc = lib.hiprec_mf_bpr_fused_step(None, None, None, None, 4, 0, 0.25, None, None)
    assert rc == -1 and b"NULL step" in lib.hiprec_last_error()
    assert lib.hiprec_fused_step_bytes() == ctypes.sizeof(_lib.FusedStep)
    assert lib.hiprec_dp_step_bytes() == ctypes.sizeof(_lib.DpStep)
    rc = lib.hiprec_mf_dp_step_begin(None, None, None, None, 4, 0.25, None)
    assert rc == -1 and b"NULL step context" in lib.hiprec_last_error()
    rc = lib.hiprec_mf_dp_step_end(None, None)
    assert rc == -1

    rc = lib.hiprec_shard_route_triples(None, None, None, 8, 65, 16, None, None, None, None)
    assert rc == -1 and b"bad routing sizes" in lib.hiprec_last_error()
    rc = lib.hiprec_shard_route_triples(None, None, None, 8, 2, 16, None, None, None, None)
    assert rc == -1 and b"NULL" in lib.hiprec_last_error()
    rc = lib.hiprec_shard_split_rows(None, 8, 0, None, None, None)
    assert rc == -1 and b"bad sizes" in lib.hiprec_last_error()
    rc = lib.hiprec_random_permutation(None, -1, 3, None)
    assert rc == -1 and b"bad permutation" in lib.hiprec_last_error()


def test_round4_entry_points_validate_before_touching_the_gpu():
    """The lazy optimizer, the staging helpers of big batches, the bucketed ownership tables, the LightGCN optimizer +
    staging launch and the planned-steps driver's up-front plan check: bad arguments come back as error codes with a
    message, nothing is launched."""
    from beta_recsys_amd import _lib

    lib = _lib.load()
    assert lib.hiprec_lazy_state_bytes() == ctypes.sizeof(_lib.LazyState)
    assert lib.hiprec_shard_bufs_bytes() == ctypes.sizeof(_lib.ShardBufs)
    st = _lib.LazyState(0, 0, 0, 0, 10, 5, 8, _lib.HIPREC_OPT_SGD if hasattr(_lib, "HIPREC_OPT_SGD") else 0, 0, 0, 0, 0, 0,
                        0.05, 0.9, 0.999, 1e-8)
    rows = _lib.LazyRows(None, 0, None, 0, None, 0, None, 0)
    rc = lib.hiprec_lazy_update(ctypes.byref(st), ctypes.byref(rows), None, None, None)
    assert rc == -1 and b"NULL pointer" in lib.hiprec_last_error()
    stats = (ctypes.c_char * 128)()
    rc = lib.hiprec_lazy_update(ctypes.byref(st), ctypes.byref(rows), None, stats, None)
    assert rc == -1 and b"Adam and RMSprop" in lib.hiprec_last_error()
    st.kind = 1
    rc = lib.hiprec_lazy_catchup(ctypes.byref(st), ctypes.byref(rows), stats, None)
    assert rc == -1 and b"NULL buffer in the lazy optimizer state" in lib.hiprec_last_error()
    rc = lib.hiprec_lazy_flush(None, stats, None)
    assert rc == -1
    rc = lib.hiprec_mf_epoch_lazy(None, None, None, None, None, None, 0, 4, 2, 1, 0.0, None, None, 0, None)
    assert rc == -1 and b"NULL pointer" in lib.hiprec_last_error()
    # ... and its owned-gradient form: the state, then the batch / ownership arrays
    rc = lib.hiprec_mf_epoch_lazy_owned(None, None, None, None, None, None, None, None, 0, 4, 2, 1, 0.0, None, None, None)
    assert rc == -1 and b"NULL pointer" in lib.hiprec_last_error()
    scratch = (ctypes.c_char * 64)()
    rc = lib.hiprec_mf_epoch_lazy_owned(ctypes.byref(st), None, None, None, None, None, None, None, 0, 4, 2, 1, 0.0,
                                        stats, scratch, None)
    assert rc == -1 and b"NULL batch / ownership arrays" in lib.hiprec_last_error()
    assert lib.hiprec_mf_epoch_lazy_owned(ctypes.byref(st), None, None, None, None, None, None, None, 0, 0, 2, 0, 0.0,
                                          stats, scratch, None) == 0      # an empty epoch that does not reset sums
    rc = lib.hiprec_mf_bpr_grad_owned(None, None, 10, 5, 8, None, None, None, None, None, None, None, 4, 0.25, 0.0, None,
                                      None, None)
    assert rc == -1 and b"NULL pointer" in lib.hiprec_last_error()
    w = (ctypes.c_float * 4)()
    rc = lib.hiprec_mf_bpr_grad_owned(w, w, 10, 5, 300, None, None, None, None, None, None, None, 4, 0.25, 0.0, stats,
                                      scratch, None)
    assert rc == -1 and b"dim <= 256" in lib.hiprec_last_error()
    rc = lib.hiprec_mf_bpr_grad_owned(w, w, 10, 5, 8, None, None, None, None, None, None, None, 4, 0.25, 0.0, stats,
                                      scratch, None)
    assert rc == -1 and b"NULL index / ownership arrays" in lib.hiprec_last_error()
    assert lib.hiprec_mf_bpr_grad_owned(w, w, 10, 5, 8, None, None, None, None, None, None, None, 0, 0.25, 0.0, stats,
                                        scratch, None) == 0              # an empty batch launches nothing
    # clip + sweep in two launches: the clip's checks come first
    rc = lib.hiprec_clip_opt_dense_step(1, None, None, None, None, 5, 1e-3, 0.9, 0.999, 1e-8, None, None, -1, 1.0, None, 0,
                                        None)
    assert rc == -1 and b"NULL pointer" in lib.hiprec_last_error()
    rc = lib.hiprec_clip_opt_dense_step(1, None, w, None, None, 4, 1e-3, 0.9, 0.999, 1e-8, None, None, -1, 1.0, scratch, 8,
                                        None)
    assert rc == -1 and b"workspace" in lib.hiprec_last_error()

    rc = lib.hiprec_stage_sort_keys(None, None, 0, 0, 8, 4, 10, 3, None, None)
    assert rc == -1 and b"bad sizes" in lib.hiprec_last_error()
    rc = lib.hiprec_stage_sort_keys(None, None, 0, 0, 8, 1, 2**30, 4, None, None)
    assert rc == -1                                     # 8 batches x 2^30 items do not fit 32-bit keys
    rc = lib.hiprec_gather_epoch(None, None, None, None, 0, 0, None, 5, None, None, None, None)
    assert rc == -1 and b"NULL pointer" in lib.hiprec_last_error()
    assert lib.hiprec_gather_epoch(None, None, None, None, 0, 0, None, 0, None, None, None, None) == 0   # nothing to do
    rc = lib.hiprec_group_epoch_by_item(None, None, None, 8, 4, 10, 6, None, None, None, None, None, None, None, None,
                                        None, None)
    assert rc == -1 and b"NULL pointer" in lib.hiprec_last_error()

    bits = lib.hiprec_ownership_table_bits(65536)
    assert bits == 18 and lib.hiprec_ownership_table_bits(1 << 30) == 24
    assert lib.hiprec_ownership_ws_ints(3 * 65536, 65536, bits) == 6 * 3 * 65536 + 3 * 33 + 3 * 64
    assert lib.hiprec_ownership_ws_ints(10, 4, 30) == 0
    rc = lib.hiprec_batch_row_ownership(None, None, None, 8, 1 << 23, 10, 10, 26, None, None, None, None)
    assert rc == -1 and b"at most 2^24" in lib.hiprec_last_error()

    rc = lib.hiprec_lightgcn_opt_stage(None, 1, None, None, None, 0.05, 0.9, 0.999, 1e-8, None, None, 0.6, 0, 1, None)
    assert rc == -1 and b"NULL plan" in lib.hiprec_last_error()

    # the planned-steps driver refuses a plan that does not add up BEFORE it posts anything
    import numpy as np

    one = np.ones(4, dtype=np.int64)
    plan = _lib.ShardPlan(1, 0, 1, 4, 4, 4, 1, 1, 1, 1, 1, 64, 1, 1, 1, np.array([0, 3], dtype=np.int64).ctypes.data,
                          one.ctypes.data, np.array([2], dtype=np.int64).ctypes.data, np.array([1], dtype=np.int64).ctypes.data,
                          None, 0, None, 0)
    bufs = _lib.ShardBufs(1, 4, 4, 8, 0, 1, 1, 1, 1, 1, 1, 1, None, None, None, None, None, None, 0)
    rc = lib.hiprec_shard_planned_steps(ctypes.byref(plan), ctypes.byref(bufs), 0, 1, 0, 0.0, 0.05, 0.9, 0.999, 1e-8, None,
                                        None, stats, None)
    assert rc == -1 and b"inconsistent plan" in lib.hiprec_last_error()


def test_mf_initial_weights_match_reference_for_same_seed():
    """MF.__init__ consumes the torch RNG like models/mf.py:21-30 -> bit-identical init."""
    g = load_golden("mf_init")
    for tag in ("a", "b"):
        U, I, D, seed = (int(x) for x in g[f"{tag}/meta"])
        torch.manual_seed(seed)
        eng = make_engine(U, I, D)
        sd = eng.model.state_dict()
        assert list(sd.keys()) == list(KEYS)
        for k in KEYS:
            assert tuple(sd[k].shape) == g[f"{tag}/w/{k}"].shape
            assert np.array_equal(sd[k].numpy(), g[f"{tag}/w/{k}"]), f"{tag} {k} differs"


def test_parameters_are_views_of_one_flat_buffer():
    eng = make_engine(6, 5, 4)
    m = eng.model
    U, I, D = 6, 5, 4
    assert m.flat.numel() == (U + I) * (D + 1) + 1
    assert m.user_emb.weight.data_ptr() == m.flat.data_ptr()
    assert m.item_emb.weight.data_ptr() == m.flat.data_ptr() + 4 * U * D
    assert m.global_bias.data_ptr() == m.flat.data_ptr() + 4 * ((U + I) * (D + 1))
    # load_state_dict copies in place: the views stay attached
    sd = {k: torch.full_like(v, 0.5) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    assert float(m.flat.min()) == 0.5 and float(m.flat.max()) == 0.5
    # .to() re-binds every view to the moved buffer
    m.to(torch.device("cpu"))
    m.flat[0] = 7.0
    assert float(m.user_emb.weight[0, 0]) == 7.0
    with pytest.raises(TypeError):
        m.half()
    t = m.tables()
    assert (t.n_users, t.n_items, t.dim) == (U, I, D) and t.user_emb == m.flat.data_ptr()


def test_engine_surface_and_quirks():
    eng = make_engine(optimizer="adam")
    for attr in ("model", "optimizer", "device", "writer", "config", "batch_size", "loss", "reg"):
        assert hasattr(eng, attr)
    assert eng.loss == "bpr" and eng.device == torch.device("cpu")
    assert eng.optimizer.defaults["betas"] == (0.9, 0.999) and eng.optimizer.defaults["eps"] == 1e-8
    # quirk Q1: model-level reg is ignored unless the key also exists at the top level
    cfg_model_reg = make_engine()
    cfg_model_reg.config["model"]["reg"] = 0.001
    assert make_engine().reg == 0.0
    import beta_recsys_amd as hp
    cfg = {"model": dict(n_users=3, n_items=3, emb_dim=2, device_str="cpu", optimizer="sgd", lr=0.1,
                         batch_size=2, reg=0.001), "system": {"run_dir": "/tmp/x"}}
    with contextlib.redirect_stdout(io.StringIO()):
        assert hp.MFEngine(cfg).reg == 0.0
        cfg["reg"] = 1
        assert hp.MFEngine(cfg).reg == 0.001
    # losses helpers mirror torch_engine.py:92-121
    p, n = torch.tensor([0.7, 0.2]), torch.tensor([0.1, 0.4])
    assert torch.allclose(eng.bpr_loss(p, n), -torch.nn.functional.logsigmoid(p - n).mean())
    assert torch.allclose(eng.bce_loss(p, torch.tensor([1.0, 0.0])),
                          torch.nn.BCELoss()(p, torch.tensor([1.0, 0.0])))


def test_unknown_optimizer_and_no_cpu_fallback():
    with pytest.raises(ValueError, match="Unsupported optimizer"):
        make_engine(optimizer="adagrad")
    eng = make_engine()
    batch = (torch.tensor([0, 1]), torch.tensor([0, 1]), torch.tensor([1, 2]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        eng.train_single_batch(batch)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        eng.model.predict(np.array([0]), np.array([0]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        eng.train_an_epoch([batch], 0)
    eng.loss = "hinge"
    with pytest.raises(RuntimeError, match="Unsupported loss type"):
        eng.train_single_batch(batch)
    eng.loss = "bpr"
    with pytest.raises(IndexError):  # quirk Q4: a batch of one cannot be trained on
        eng.train_single_batch((torch.tensor([0]), torch.tensor([0]), torch.tensor([1])))


def test_checkpoint_format(tmp_path):
    """torch.save(state_dict) with the reference's keys / shapes (torch_engine.py:70-90)."""
    eng = make_engine(9, 8, 6)
    path = str(tmp_path / "mf.model")
    eng.save_checkpoint(path)
    sd = torch.load(path)
    assert list(sd.keys()) == list(KEYS)
    assert sd["user_emb.weight"].shape == (9, 6) and sd["item_bias.weight"].shape == (8, 1)
    assert sd["global_bias"].shape == (1,)
    other = make_engine(9, 8, 6)
    with contextlib.redirect_stdout(io.StringIO()):
        model = other.resume_checkpoint(path)
    assert model is other.model
    for k in KEYS:
        assert torch.equal(other.model.state_dict()[k], sd[k])
    # a stock torch module with the reference's layout can load it, and vice versa
    ref_like = torch.nn.ModuleDict({"user_emb": torch.nn.Embedding(9, 6), "item_emb": torch.nn.Embedding(8, 6),
                                    "user_bias": torch.nn.Embedding(9, 1), "item_bias": torch.nn.Embedding(8, 1)})
    ref_like.global_bias = torch.nn.Parameter(torch.zeros(1))
    ref_like.load_state_dict(sd)
    other.model.load_state_dict(ref_like.state_dict())


def test_device_triple_batcher_composition():
    import beta_recsys_amd as hp

    n, bs = 103, 16
    u = torch.arange(n)
    b = hp.DeviceTripleBatcher(u, u + 1000, u + 2000, bs, generator=torch.Generator().manual_seed(1))
    batches = list(b)
    assert len(b) == len(batches) == 7
    assert [len(x[0]) for x in batches] == [16] * 6 + [7]  # drop_last=False
    seen = torch.cat([x[0] for x in batches])
    assert sorted(seen.tolist()) == list(range(n))  # a permutation: every triple exactly once
    assert seen.tolist() != list(range(n))
    for x in batches:
        assert torch.equal(x[1], x[0] + 1000) and torch.equal(x[2], x[0] + 2000)
        assert all(t.dtype == torch.int64 for t in x)
    seq = hp.DeviceTripleBatcher(u, u, u, bs, shuffle=False)
    assert seq.permutation() is None
    assert torch.cat([x[0] for x in seq]).tolist() == list(range(n))
    with pytest.raises(ValueError):
        hp.DeviceTripleBatcher(u, u[:-1], u, bs)


def test_compat_install_routes_reference_imports(tmp_path, monkeypatch):
    """A package laid out like beta_rec resolves `from ..models.mf import MFEngine` to the mirror."""
    import importlib
    import sys

    import beta_recsys_amd as hp
    from beta_recsys_amd import compat

    root = tmp_path / "fake"
    (root / "beta_rec" / "models").mkdir(parents=True)
    (root / "beta_rec" / "recommenders").mkdir(parents=True)
    (root / "beta_rec" / "__init__.py").write_text("")
    (root / "beta_rec" / "models" / "__init__.py").write_text("")
    (root / "beta_rec" / "models" / "mf.py").write_text("raise ImportError('the reference module must not load')\n")
    # the reference's own base module stays the reference's (~15 un-mirrored engines -- vbcar,
    # narm, sasrec, cmn ... -- subclass it and call self.optimizer.step())
    (root / "beta_rec" / "models" / "torch_engine.py").write_text("class ModelEngine:\n    reference = True\n")
    (root / "beta_rec" / "recommenders" / "__init__.py").write_text("")
    (root / "beta_rec" / "recommenders" / "matrix_factorization.py").write_text(
        "from ..models.mf import MFEngine\nfrom beta_rec.models.torch_engine import ModelEngine\n")
    # the other call sites the mirrors serve: recommenders/ngcf.py:11, recommenders/triple2vec.py:9,
    # examples/train_cmn.py:13, models/cmn.py:7 (truncated_normal_)
    (root / "beta_rec" / "recommenders" / "siblings.py").write_text(
        "from ..models.ngcf import NGCFEngine\nfrom ..models.triple2vec import Triple2vecEngine\n"
        "from beta_rec.models.pairwise_gmf import PairwiseGMFEngine, truncated_normal_\n"
        "from ..models.lightgcn import LightGCNEngine\nfrom ..models.ncf import NeuMFEngine\n")
    monkeypatch.syspath_prepend(str(root))
    saved = {k: v for k, v in sys.modules.items() if k == "beta_rec" or k.startswith("beta_rec.")}
    for k in saved:
        del sys.modules[k]
    try:
        installed = compat.install()
        assert installed[0] == "beta_rec.models.mf" and "beta_rec.models.torch_engine" not in installed
        m = importlib.import_module("beta_rec.recommenders.matrix_factorization")
        assert m.MFEngine is hp.MFEngine
        assert m.ModelEngine is not hp.ModelEngine and m.ModelEngine.reference
        sib = importlib.import_module("beta_rec.recommenders.siblings")
        assert sib.NGCFEngine is hp.NGCFEngine and sib.Triple2vecEngine is hp.Triple2vecEngine
        assert sib.PairwiseGMFEngine is hp.PairwiseGMFEngine and callable(sib.truncated_normal_)
        assert sib.LightGCNEngine is hp.LightGCNEngine and sib.NeuMFEngine is hp.NeuMFEngine
        models_pkg = sys.modules["beta_rec.models"]
        compat.install()  # idempotent; now that the parent package is loaded it also gets the attributes
        assert models_pkg.mf is sys.modules["beta_rec.models.mf"]
        # opt-in: the HIP base under the reference's name
        compat.install(extra={"beta_rec.models.torch_engine": "torch_engine"})
        assert sys.modules["beta_rec.models.torch_engine"].ModelEngine is hp.ModelEngine
        compat.uninstall()
        # uninstall puts back what was there: the reference's torch_engine module, and no mirror
        # left behind either in sys.modules or as an attribute of the parent package
        assert sys.modules["beta_rec.models.torch_engine"].ModelEngine.reference
        assert models_pkg.torch_engine.ModelEngine.reference
        assert "beta_rec.models.mf" not in sys.modules and not hasattr(models_pkg, "mf")
    finally:
        compat.uninstall()
        for k in [k for k in sys.modules if k.startswith("beta_rec.") or k == "beta_rec"]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_model_engine_builds_torch_optimizer_for_foreign_models():
    """A reference engine that is NOT mirrored but subclasses the HIP ``ModelEngine`` (compat.install with the
    torch_engine extra) keeps the torch optimizer torch_engine.py:23-39 builds: optimizer.step() works."""
    import beta_recsys_amd as hp

    class Foreign(hp.ModelEngine):
        def __init__(self, config):
            self.model = torch.nn.Linear(3, 1)
            super().__init__(config)

    for name, cls in (("sgd", torch.optim.SGD), ("adam", torch.optim.Adam), ("rmsprop", torch.optim.RMSprop)):
        with contextlib.redirect_stdout(io.StringIO()):
            eng = Foreign({"model": {"optimizer": name, "lr": 0.1, "device_str": "cpu"},
                           "system": {"run_dir": "/tmp/hiprec_test_runs"}})
        assert isinstance(eng.optimizer, cls)
        before = eng.model.weight.detach().clone()
        eng.optimizer.zero_grad()
        eng.model(torch.ones(2, 3)).sum().backward()
        eng.optimizer.step()
        assert not torch.equal(before, eng.model.weight)
    with pytest.raises(ValueError):
        with contextlib.redirect_stdout(io.StringIO()):
            Foreign({"model": {"optimizer": "lbfgs", "lr": 0.1, "device_str": "cpu"},
                     "system": {"run_dir": "/tmp/hiprec_test_runs"}})


def ncf_config(U, I, E, L, optimizer="adam", lr=1e-3, B=8, device="cpu", model="ncf_end"):
    return {"model": dict(n_users=U, n_items=I, emb_dim=E, dropout=0.0, device_str=device,
                          optimizer=optimizer, lr=lr, batch_size=B, model=model,
                          mlp_config={"n_layers": L, "name": "mlp", "save_name": "mlp.model"},
                          gmf_config={"name": "gmf", "save_name": "gmf.model"}),
            "system": {"run_dir": "/tmp/hiprec_test_runs", "model_save_dir": "/tmp/hiprec_test_runs"}}


@pytest.mark.parametrize("case,engine", [("ncf_neumf_adam", "NeuMFEngine"), ("ncf_gmf_adam", "GMFEngine"),
                                         ("ncf_mlp_adam", "MLPEngine"), ("ncf_neumf_sgd_e32", "NeuMFEngine")])
def test_ncf_family_init_and_state_dict_match_reference(case, engine):
    """Same torch seed -> bit-identical initial weights (incl. quirk Q8), same state_dict keys."""
    import beta_recsys_amd as hp

    g = load_golden(case)
    U, I, E, L, B, _, seed = (int(x) for x in g["meta"])
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = getattr(hp, engine)(ncf_config(U, I, E, L, str(g["optimizer"]), float(g["lr"]), B))
    sd = eng.model.state_dict()
    ref_keys = [k[3:] for k in g if k.startswith("w0/")]
    assert list(sd.keys()) == ref_keys
    for k in ref_keys:
        assert tuple(sd[k].shape) == g[f"w0/{k}"].shape, k
        assert np.array_equal(sd[k].numpy(), g[f"w0/{k}"]), f"{k} differs from the reference init"
    # views of one flat buffer, .to() keeps them attached
    m = eng.model
    assert m.flat.numel() == sum(v.numel() for v in sd.values())
    m.to(torch.device("cpu"))
    first = next(iter(sd))
    m.flat[0] = 3.0
    assert float(m.state_dict()[first].reshape(-1)[0]) == 3.0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        eng.train_single_batch(torch.tensor([0, 1]), torch.tensor([0, 1]), torch.tensor([1.0, 0.0]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.predict(np.array([0]), np.array([0]))


def test_lightgcn_init_and_surface_on_cpu():
    """Same seed -> the reference's xavier init; reference state_dict keys; no CPU compute path."""

    import beta_recsys_amd as hp

    g = load_golden("lightgcn_adam")
    U, I, D, L, B, _, seed = (int(x) for x in g["meta"])
    N = U + I
    idx = torch.from_numpy(np.vstack((g["adj_row"], g["adj_col"])).astype(np.int64))
    norm = torch.sparse_coo_tensor(idx, torch.from_numpy(g["adj_val"]), torch.Size((N, N)))
    torch.manual_seed(seed)
    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=D, layer_size=[D] * L, keep_pro=0.6, regs=[1e-5],
                         device_str="cpu", optimizer="adam", lr=0.05, batch_size=B, norm_adj=norm),
           "system": {"run_dir": "/tmp/hiprec_test_runs"}}
    with contextlib.redirect_stdout(io.StringIO()):
        eng = hp.LightGCNEngine(cfg)
    sd = eng.model.state_dict()
    assert list(sd.keys()) == ["user_embedding.weight", "item_embedding.weight"]
    for k in sd:
        assert np.array_equal(sd[k].numpy(), g[f"w0/{k}"]), k
    assert eng.decay == 1e-5 and eng.model.n_layers == L
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        eng.train_single_batch((torch.tensor([0, 1]), torch.tensor([0, 1]), torch.tensor([1, 2])))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        eng.model.predict(np.array([0]), np.array([0]))


def test_ncf_pre_merges_gmf_and_mlp_checkpoints(tmp_path):
    """models/ncf.py:155-193: `ncf_pre` builds NeuMF from trained GMF + MLP checkpoints —
    embeddings copied, tower copied, affine_output = 0.5 * cat(mlp, gmf), bias = 0.5 * (sum)."""
    import beta_recsys_amd as hp

    U, I, E, L = 13, 11, 4, 2
    cfg = ncf_config(U, I, E, L)
    cfg["system"]["model_save_dir"] = str(tmp_path)
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        gmf = hp.GMFEngine(cfg)
        mlp = hp.MLPEngine(cfg)
    gmf.save_checkpoint(str(tmp_path / "gmf.model"))
    mlp.save_checkpoint(str(tmp_path / "mlp.model"))
    pre = ncf_config(U, I, E, L, model="ncf_pre")
    pre["system"]["model_save_dir"] = str(tmp_path)
    with contextlib.redirect_stdout(io.StringIO()):
        neu = hp.NeuMFEngine(pre)
    n, g, m = neu.model.state_dict(), gmf.model.state_dict(), mlp.model.state_dict()
    assert torch.equal(n["embedding_user_mf.weight"], g["embedding_user.weight"])
    assert torch.equal(n["embedding_item_mf.weight"], g["embedding_item.weight"])
    assert torch.equal(n["embedding_user_mlp.weight"], m["embedding_user.weight"])
    assert torch.equal(n["embedding_item_mlp.weight"], m["embedding_item.weight"])
    for k in ("fc_layers.1.weight", "fc_layers.1.bias", "fc_layers.4.weight", "fc_layers.4.bias"):
        assert torch.equal(n[k], m[k])
    assert torch.equal(n["affine_output.weight"],
                       0.5 * torch.cat([m["affine_output.weight"], g["affine_output.weight"]], dim=-1))
    assert torch.equal(n["affine_output.bias"], 0.5 * (m["affine_output.bias"] + g["affine_output.bias"]))
    # the merged parameters are still views of NeuMF's single flat buffer
    assert neu.model.embedding_user_mlp.weight.data_ptr() == neu.model.flat.data_ptr()


def test_compat_install_eval_patches_evaluate_and_keeps_rating_metrics(tmp_path, monkeypatch):
    """install_eval swaps eval_engine.evaluate for the HIP one; non-ranking metrics stay with the reference."""
    import importlib
    import sys

    from beta_recsys_amd import compat
    from beta_recsys_amd import eval as hip_eval

    root = tmp_path / "fake"
    (root / "beta_rec" / "core").mkdir(parents=True)
    (root / "beta_rec" / "__init__.py").write_text("")
    (root / "beta_rec" / "core" / "__init__.py").write_text("")
    (root / "beta_rec" / "core" / "eval_engine.py").write_text(
        "def evaluate(data_df, predictions, metrics, k_li):\n    return {'from': 'reference'}\n"
        "def train_eval_worker(df, pred, metrics, k):\n    return evaluate(df, pred, metrics, k)\n")
    monkeypatch.syspath_prepend(str(root))
    saved = {k: v for k, v in sys.modules.items() if k == "beta_rec" or k.startswith("beta_rec.")}
    for k in saved:
        del sys.modules[k]
    seen = []
    monkeypatch.setattr(hip_eval, "evaluate", lambda *a, **k: seen.append(a) or {"from": "hip"})
    try:
        compat.install_eval()
        ee = importlib.import_module("beta_rec.core.eval_engine")
        assert compat.install_eval() is ee.evaluate          # idempotent
        assert ee.train_eval_worker(None, None, ["ndcg", "recall"], [10]) == {"from": "hip"}
        assert ee.train_eval_worker(None, None, ["rmse"], [10]) == {"from": "reference"}
        assert len(seen) == 1
        compat.uninstall_eval()
        assert ee.train_eval_worker(None, None, ["ndcg"], [10]) == {"from": "reference"}
    finally:
        for k in [k for k in sys.modules if k.startswith("beta_rec.") or k == "beta_rec"]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_eval_group_by_user_and_argument_checks():
    from beta_recsys_amd import eval as hip_eval

    order, seg = hip_eval.group_by_user(torch.tensor([4, 4, 7, 9, 9, 9]))
    assert order is None and seg.tolist() == [0, 2, 3, 6]                   # already grouped: no sort
    order, seg = hip_eval.group_by_user(torch.tensor([9, 4, 9, 7, 4, 9]))
    assert order.tolist() == [1, 4, 3, 0, 2, 5] and seg.tolist() == [0, 2, 3, 6]   # stable within a user
    order, seg = hip_eval.group_by_user(torch.tensor([], dtype=torch.int64))
    assert order is None and seg.tolist() == [0]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip_eval.rank_metrics([0], [1.0], [0.5], [1], device="cpu")
    with pytest.raises(ValueError):
        hip_eval.rank_metrics([0], [1.0], [0.5], [], device="cpu")
    with pytest.raises(KeyError):
        hip_eval.evaluate({"col_user": [0], "col_rating": [1.0]}, [0.5], ["rmse"], 5)


def test_compat_install_data_patches_the_loader_builders(tmp_path, monkeypatch):
    """install_data swaps BaseData.instance_*_loader for the device-side builders, same signatures."""
    import importlib
    import sys

    from beta_recsys_amd import compat
    from beta_recsys_amd import data as hip_data

    root = tmp_path / "fake"
    (root / "beta_rec" / "data").mkdir(parents=True)
    (root / "beta_rec" / "__init__.py").write_text("")
    (root / "beta_rec" / "data" / "__init__.py").write_text("")
    (root / "beta_rec" / "data" / "base_data.py").write_text(
        "class BaseData:\n"
        "    def instance_bpr_loader(self, batch_size, device):\n        return 'ref-bpr'\n"
        "    def instance_bce_loader(self, batch_size, device, num_negative):\n        return 'ref-bce'\n"
        "    def instance_mul_neg_loader(self, batch_size, device, num_negative):\n        return 'ref-mul'\n")
    monkeypatch.syspath_prepend(str(root))
    saved = {k: v for k, v in sys.modules.items() if k == "beta_rec" or k.startswith("beta_rec.")}
    for k in saved:
        del sys.modules[k]
    calls = []
    for name in ("instance_bpr_loader", "instance_bce_loader", "instance_mul_neg_loader"):
        monkeypatch.setattr(hip_data, name, lambda data, *a, _n=name, **k: calls.append((_n, data, a, k)) or _n)
    try:
        base = compat.install_data()
        compat.install_data()                                    # idempotent
        obj = importlib.import_module("beta_rec.data.base_data").BaseData()
        assert obj.instance_bpr_loader(512, "cuda:0") == "instance_bpr_loader"
        assert obj.instance_bce_loader(512, "cuda:0", 4) == "instance_bce_loader"
        assert obj.instance_mul_neg_loader(512, "cuda:0", num_negative=4) == "instance_mul_neg_loader"
        assert calls[0] == ("instance_bpr_loader", obj, (512, "cuda:0"), {})
        assert calls[2][3] == {"num_negative": 4}
        compat.uninstall_data()
        assert base().instance_bpr_loader(1, "cpu") == "ref-bpr"
    finally:
        for k in [k for k in sys.modules if k.startswith("beta_rec.") or k == "beta_rec"]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_positive_csr_and_loader_argument_checks():
    from beta_recsys_amd import data as hip_data

    users = torch.tensor([2, 0, 2, 2, 0, 3])
    items = torch.tensor([5, 1, 0, 5, 4, 2])                      # (2, 5) twice: a set per user
    ptr, cols = hip_data.build_positive_csr(users, items, 5, 6)
    assert ptr.tolist() == [0, 2, 2, 4, 5, 5] and cols.tolist() == [1, 4, 0, 5, 2]
    with pytest.raises(IndexError):
        hip_data.build_positive_csr(users, items, 3, 6)           # user 3 outside [0, 3)
    with pytest.raises(ValueError):
        hip_data.sample_negatives(users, items, 5, 6, k=0, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip_data.sample_negatives(users, items, 5, 6, device="cpu")


def pgmf_config(U=9, I=7, D=4, **model):
    m = {"device_str": "cpu", "optimizer": "adam", "lr": 1e-3}
    m.update(model)
    return {"n_users": U, "n_items": I, "emb_dim": D, "regs": [1e-5], "batch_size": 8, "lr": 1e-4,
            "pretrain_l2_lambda": 1e-4, "grad_clip": 5.0, "neg_count": 4, "model": m,
            "system": {"run_dir": "/tmp/hiprec_test_runs"}}


def test_pairwise_gmf_initial_weights_match_reference_for_same_seed():
    """PairwiseGMF.__init__ consumes the torch RNG like models/pairwise_gmf.py:35-46."""
    import beta_recsys_amd as hp

    g = load_golden("pgmf_init")
    for tag in ("a", "b"):
        U, I, D, seed = (int(x) for x in g[f"{tag}/meta"])
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            eng = hp.PairwiseGMFEngine(pgmf_config(U, I, D))
        sd = eng.model.state_dict()
        assert list(sd.keys()) == ["user_memory.weight", "item_memory.weight", "v.weight"]
        for k in sd:
            assert np.array_equal(sd[k].numpy(), g[f"{tag}/w/{k}"]), f"{tag} {k} differs"


def test_pairwise_gmf_engine_surface():
    import beta_recsys_amd as hp
    from beta_recsys_amd import _lib, compat

    with contextlib.redirect_stdout(io.StringIO()):
        eng = hp.PairwiseGMFEngine(pgmf_config())
        keeps_adam = hp.PairwiseGMFEngine(pgmf_config(optimizer="none-of-the-three"))
        sgd = hp.PairwiseGMFEngine(pgmf_config(optimizer="sgd", lr=0.5))
    # torch_engine.py:23-39 replaces the engine's Adam(lr=config["lr"]) only for a known name
    assert (eng.optimizer.name, eng.optimizer.lr) == ("adam", 1e-3)
    assert (keeps_adam.optimizer.name, keeps_adam.optimizer.lr) == ("adam", 1e-4)
    assert (sgd.optimizer.name, sgd.optimizer.lr) == ("sgd", 0.5)
    assert eng.batch_size == 8 and eng.regs == [1e-5]
    m = eng.model
    t = m.tables()
    assert t.user_memory == m.flat.data_ptr() and t.v == m.flat.data_ptr() + 4 * (9 + 7) * 4
    assert m.flat.numel() == (9 + 7 + 1) * 4 and m.v.weight.shape == (1, 4)
    p, n = torch.tensor([[0.7], [0.2]]), torch.tensor([[0.1], [0.4]])
    assert torch.allclose(eng.bpr_loss(p, n), (-torch.log(torch.sigmoid(p - n) + 1e-12)).mean())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        eng.train_single_batch(([0, 1], [1, 2], [3, 4]))
    assert compat.MIRRORS["beta_rec.models.pairwise_gmf"] == "pairwise_gmf"
    # C-ABI argument checks fire before any launch
    lib = _lib.load()
    assert lib.hiprec_pgmf_workspace_bytes(64) >= 64 * 4
    rc = lib.hiprec_pgmf_bpr_grad(None, None, None, None, None, 4, 0.25, 0.0, None, None, 0, None, 0, None)
    assert rc == -1 and b"NULL tables" in lib.hiprec_last_error()
    big = _lib.PgmfTables(8, 8, 8, 3, 3, 300, 0)
    rc = lib.hiprec_pgmf_bpr_grad(ctypes.byref(big), ctypes.byref(big), None, None, None, 0, 0.25, 0.0, None,
                                  None, 0, None, 0, None)
    assert rc == -1 and b"dim <= 256" in lib.hiprec_last_error()
    rc = lib.hiprec_clip_grad_norm(None, 5, 1.0, None, 0, None)
    assert rc == -1 and b"NULL" in lib.hiprec_last_error()
    rc = lib.hiprec_pgmf_epoch(None, None, None, None, None, 10, 0, 0.0, 5.0, 1, 1e-3, 0.9, 0.999, 1e-8, None, None,
                               None, None, 0, None, None, 0, None, 0, None, 0, None)
    assert rc == -1 and b"bad n_triples/batch" in lib.hiprec_last_error()
    rc = lib.hiprec_pgmf_epoch(None, None, None, None, None, 10, 4, 0.0, 5.0, 1, 1e-3, 0.9, 0.999, 1e-8, None, None,
                               None, None, 0, None, None, 0, None, 0, None, 0, None)
    assert rc == -1 and b"flat buffers" in lib.hiprec_last_error()


def t2v_config(U=9, I=7, D=4, n_neg=2, **model):
    m = dict(n_users=U, n_items=I, emb_dim=D, n_neg=n_neg, batch_size=8, device_str="cpu", optimizer="adam",
             lr=1e-3)
    m.update(model)
    return {"model": m, "system": {"run_dir": "/tmp/hiprec_test_runs"}}


def test_triple2vec_initial_weights_match_reference_for_same_seed():
    """Triple2vec.__init__ consumes the torch RNG like models/triple2vec.py:21-34."""
    import beta_recsys_amd as hp

    g = load_golden("t2v_init")
    for tag in ("a", "b"):
        U, I, D, seed = (int(x) for x in g[f"{tag}/meta"])
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            eng = hp.Triple2vecEngine(t2v_config(U, I, D))
        sd = eng.model.state_dict()
        assert list(sd.keys()) == ["user_emb.weight", "item_emb1.weight", "item_emb2.weight",
                                   "user_bias.weight", "item_bias.weight"]
        for k in sd:
            assert np.array_equal(sd[k].numpy(), g[f"{tag}/w/{k}"]), f"{tag} {k} differs"


def test_triple2vec_item_emb2_alias_and_layout():
    """triple2vec.py:19,38-39: item_emb2 becomes item_emb1 on the first forward when n_neg != 0; the
    orphaned table sits last in the flat buffer and drops out of the optimizer sweep."""
    import beta_recsys_amd as hp
    from beta_recsys_amd import _lib, compat

    with contextlib.redirect_stdout(io.StringIO()):
        eng = hp.Triple2vecEngine(t2v_config())
    m = eng.model
    U, I, D = 9, 7, 4
    assert m.flat.numel() == U * D + 2 * I * D + U + I and m.n_active() == m.flat.numel()
    t = m.tables()
    assert t.item_emb2 == m.flat.data_ptr() + 4 * (U * D + I * D + U + I) and t.item_emb2 != t.item_emb1
    before = {k: v.clone() for k, v in m.state_dict().items()}
    assert not torch.equal(before["item_emb1.weight"], before["item_emb2.weight"])
    m._alias()
    assert m.shared_items and m.n_active() == U * D + I * D + U + I
    t = m.tables()
    assert t.item_emb2 == t.item_emb1 == m.flat.data_ptr() + 4 * U * D
    sd = m.state_dict()
    assert list(sd.keys()) == list(before.keys())
    assert torch.equal(sd["item_emb2.weight"], before["item_emb1.weight"])
    # .to() / load_state_dict keep working on the aliased model
    m.to(torch.device("cpu"))
    m.flat[U * D] = 3.0
    assert float(m.item_emb1.weight[0, 0]) == 3.0 and float(m.item_emb2.weight[0, 0]) == 3.0
    m.load_state_dict(before)
    assert torch.equal(m.item_emb1.weight, before["item_emb2.weight"])      # shared parameter: last key wins
    with contextlib.redirect_stdout(io.StringIO()):
        no_neg = hp.Triple2vecEngine(t2v_config(n_neg=0))
    no_neg.model._alias()
    assert not no_neg.model.shared_items
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        eng.train_single_batch(([0], [1], [2], [[1, 2]], [[1, 2]], [[3, 4]]))
    assert compat.MIRRORS["beta_rec.models.triple2vec"] == "triple2vec"
    lib = _lib.load()
    rc = lib.hiprec_t2v_grad(None, None, None, None, None, None, None, None, 4, 2, 0.1, None, None, 0, None)
    assert rc == -1 and b"w is NULL" in lib.hiprec_last_error()
    mixed_w, plain_g = _lib.T2vTables(8, 16, 16, 8, 8, 3, 3, 4, 0), _lib.T2vTables(8, 16, 24, 8, 8, 3, 3, 4, 0)
    rc = lib.hiprec_t2v_grad(ctypes.byref(mixed_w), ctypes.byref(plain_g), None, None, None, None, None, None, 0, 2,
                             0.1, None, None, 0, None)
    assert rc == -1 and b"alias" in lib.hiprec_last_error()
    rc = lib.hiprec_t2v_epoch(None, None, None, None, None, None, None, None, 10, 4, -1, 0.1, 1, 1e-3, 0.9, 0.999,
                              1e-8, None, None, None, None, 0, None, None, 0, None)
    assert rc == -1 and b"bad n_triples/batch/n_neg" in lib.hiprec_last_error()
    rc = lib.hiprec_alias_sample(None, None, None, 0, 1, None, 5, None)
    assert rc == -1 and b"bad sizes" in lib.hiprec_last_error()
    rc = lib.hiprec_t2v_predict(ctypes.byref(mixed_w), None, None, 3, None, None, None)
    assert rc == -1 and b"NULL pointer" in lib.hiprec_last_error()


def test_ngcf_initial_weights_and_state_dict_order_match_reference():
    """NGCF.__init__ consumes the torch RNG like models/ngcf.py:29-46; state_dict keeps its key order."""
    import beta_recsys_amd as hp
    from beta_recsys_amd import _lib, compat

    for case in ("ngcf_adam", "ngcf_sgd_widths"):
        g = load_golden(case)
        U, I, D, L, B, _, seed = (int(x) for x in g["meta"])
        idx = torch.from_numpy(np.vstack((g["adj_row"], g["adj_col"])).astype(np.int64))
        adj = torch.sparse_coo_tensor(idx, torch.from_numpy(g["adj_val"]), torch.Size((U + I, U + I)))
        model = dict(n_users=U, n_items=I, emb_dim=D, layer_size=[int(x) for x in g["layers"]],
                     mess_dropout=[float(x) for x in g["mess_dropout"]], regs=[1e-5], device_str="cpu",
                     optimizer=str(g["optimizer"]), lr=float(g["lr"]), batch_size=B, norm_adj=adj)
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            eng = hp.NGCFEngine({"model": model, "system": {"run_dir": "/tmp/hiprec_test_runs"}})
        sd = eng.model.state_dict()
        assert list(sd.keys()) == [k[len("init/"):] for k in g if k.startswith("init/")]
        for k in sd:
            assert np.array_equal(sd[k].numpy(), g[f"init/{k}"]), f"{case} {k} differs"
        m = eng.model
        assert m.flat.data_ptr() == m.user_embedding.weight.data_ptr()
        assert m.item_embedding.weight.data_ptr() == m.flat.data_ptr() + 4 * U * D
        assert eng.decay == 1e-5 and eng.batch_size == B
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            eng.train_single_batch(([0], [1], [2]))
    assert compat.MIRRORS["beta_rec.models.ngcf"] == "ngcf"
    lib = _lib.load()
    assert lib.hiprec_ngcf_plan_bytes() == ctypes.sizeof(_lib.NgcfPlan)
    rc = lib.hiprec_ngcf_grad(None, None, None, None, 4, 0.25, None, None, 0, None)
    assert rc == -1 and b"NULL plan" in lib.hiprec_last_error()
    plan = _lib.NgcfPlan()
    plan.n_layers = 9
    rc = lib.hiprec_ngcf_forward(ctypes.byref(plan), 0, None)
    assert rc == -1 and b"n_layers 9" in lib.hiprec_last_error()


def test_sibling_engines_checkpoint_round_trip(tmp_path):
    """save_checkpoint / resume_checkpoint (torch_engine.py:70-90) on the sibling engines: a plain
    ``torch.save(state_dict)`` with the reference's keys, loadable into a fresh engine (and, for
    Triple2vec, into one whose item_emb2 is already aliased)."""
    import beta_recsys_amd as hp

    idx = torch.tensor([[0, 1, 2, 3, 4], [1, 0, 3, 2, 4]])
    adj = torch.sparse_coo_tensor(idx, torch.ones(5), torch.Size((5, 5)))
    ngcf_model = dict(n_users=2, n_items=3, emb_dim=4, layer_size=[4, 2], mess_dropout=[0.1, 0.0], regs=[1e-5],
                      device_str="cpu", optimizer="adam", lr=1e-3, batch_size=4, norm_adj=adj)
    makers = {
        "pgmf": lambda: hp.PairwiseGMFEngine(pgmf_config()),
        "t2v": lambda: hp.Triple2vecEngine(t2v_config()),
        "ngcf": lambda: hp.NGCFEngine({"model": dict(ngcf_model), "system": {"run_dir": "/tmp/hiprec_test_runs"}}),
    }
    for name, make in makers.items():
        with contextlib.redirect_stdout(io.StringIO()):
            torch.manual_seed(1)
            a = make()
            torch.manual_seed(2)
            b = make()
        if name == "t2v":
            b.model._alias()
        path = str(tmp_path / f"{name}.model")
        a.save_checkpoint(path)
        sd = torch.load(path)
        assert list(sd.keys()) == list(a.model.state_dict().keys())
        assert not torch.equal(next(iter(b.model.state_dict().values())), next(iter(sd.values())))
        with contextlib.redirect_stdout(io.StringIO()):
            b.resume_checkpoint(path)
        for k, v in b.model.state_dict().items():
            want = sd["item_emb2.weight"] if (name == "t2v" and k.startswith("item_emb")) else sd[k]
            assert torch.equal(v, want), (name, k)          # shared parameter: the last key loaded wins
        assert b.model.flat.data_ptr() == next(iter(b.model.parameters())).data_ptr() or name != "pgmf"


def _brute_force_ownership_check(users, pos, neg, bs, U, I, own, total):
    """own / total against a per-batch count of row occurrences (python loops: small inputs only)."""
    from collections import Counter

    n = len(users)
    for b in range((n + bs - 1) // bs):
        ts = range(b * bs, min(n, (b + 1) * bs))
        ok = {t: 0 <= users[t] < U and 0 <= pos[t] < I and 0 <= neg[t] < I for t in ts}
        cu = Counter(int(users[t]) for t in ts if ok[t])
        ci = Counter([int(pos[t]) for t in ts if ok[t]] + [int(neg[t]) for t in ts if ok[t]])
        seen = {}
        for t in ts:
            if not ok[t]:
                assert own[:, t].tolist() == [-1, -1, -1]
                continue
            for role, (cnt, key) in enumerate(((cu, ("u", int(users[t]))), (ci, ("i", int(pos[t]))),
                                               (ci, ("i", int(neg[t]))))):
                c, s = cnt[key[1]], int(own[role, t])
                if c == 1:
                    assert s == -1, "a row that occurs once must be owned by its triple"
                else:
                    assert s >= 0 and int(total[b, s]) == c, (b, t, role, c, s)
                    assert seen.setdefault(s, key) == key, "two different rows share a slot"
        assert len(seen) == sum(v > 1 for v in cu.values()) + sum(v > 1 for v in ci.values())


def test_batch_row_ownership_contract():
    """The staging data of the owned-rows SGD step (csrc/mf_owned.hip), sort-based statement: a row that occurs
    once in its batch gets -1, every other row a slot of its own inside the batch with total = its occurrences
    (a positive and a negative occurrence of one item both count); triples with out-of-range ids get -1."""
    from beta_recsys_amd.mf import batch_row_ownership_torch

    rng = np.random.default_rng(0)
    n, bs, U, I = 1000, 128, 50, 30
    users, pos, neg = (torch.from_numpy(rng.integers(0, hi, n)) for hi in (U, I, I))
    users[5], pos[77], neg[500] = U + 3, -2, I
    own, total, stride = batch_row_ownership_torch(users, pos, neg, bs, U, I)
    assert stride == bs + bs // 2 and tuple(total.shape) == (8, stride) and own.dtype == torch.int32
    _brute_force_ownership_check(users.numpy(), pos.numpy(), neg.numpy(), bs, U, I, own.numpy(), total.numpy())
    own0, total0, _ = batch_row_ownership_torch(users[:0], pos[:0], neg[:0], bs, U, I)
    assert own0.shape == (3, 0)


@pytest.mark.parametrize("n,n_groups,cap,heavy,lane_slots", [(700, 16, 60, True, 16), (9746, 16, 490, True, 48),
                                                            (50, 8, 128, False, 24), (3000, 8, 100, False, 32),
                                                            (700, 16, 60, True, 48)])
def test_sliced_graph_host_covers_every_edge_once(n, n_groups, cap, heavy, lane_slots):
    """hiprec_sliced_csr as lightgcn.sliced_graph_host builds it: rows padded to lane_slots slots, every edge in exactly
    one slot of one chunk (at most 4 x lane_slots slots) of its row with its column / value / keep index, padding slots
    inert, the all-padding tail behind the rows, chunks sorted by row, subgroups = consecutive rows within the cap
    holding exactly their rows' chunks, empty rows (leading, interior, trailing) inside some subgroup."""
    from beta_recsys_amd.lightgcn import sliced_graph_host

    SLICED_PAD, SLICED_CHUNK = lane_slots, 4 * lane_slots

    rng = np.random.default_rng(n)
    lens = rng.integers(0, 200, n)
    lens[rng.random(n) < 0.2] = 0
    lens[:3] = 0
    lens[-5:] = 0
    if heavy:
        lens[7] = 5000
    rowptr = np.concatenate([[0], np.cumsum(lens)])
    nnz = int(rowptr[-1])
    col = rng.integers(0, n, nnz)
    val = rng.standard_normal(nnz).astype(np.float32)
    eid = rng.permutation(nnz).astype(np.int64)
    h = sliced_graph_host(rowptr, col, val, eid, n_groups, cap, factor=False) if lane_slots == 16 else None
    if lane_slots != 16:   # the wider geometries are for factored graphs: rank-one values
        fr, fc = rng.random(n).astype(np.float32) + 0.5, rng.random(n).astype(np.float32) + 0.5
        val = (fr[np.repeat(np.arange(n), lens)] * fc[col]).astype(np.float32)
        with pytest.raises(ValueError, match="lane_slots"):
            sliced_graph_host(rowptr, col, rng.standard_normal(nnz).astype(np.float32), eid, n_groups, cap,
                              lane_slots=lane_slots)
        h = sliced_graph_host(rowptr, col, val, eid, n_groups, cap, lane_slots=lane_slots)
        assert "col_scale" in h
    assert h is not None and h["lane_slots"] == lane_slots
    chunks, k = h["chunks"], h["subs_per_group"]
    assert k == 1
    start, row, clen = chunks[:, 0] & 0x3FFFFF, chunks[:, 1] & 0xFFFF, (chunks[:, 1] >> 16) & 0xFF
    spill_idx = (chunks[:, 0].astype(np.int64) >> 22) & 0x1FF
    real = clen > 0  # subgroups are padded with empty chunks to a multiple of 16
    assert np.all(clen[real] >= SLICED_PAD) and np.all(clen <= SLICED_CHUNK) and np.all(np.diff(row) >= 0)
    assert np.all(start % SLICED_PAD == 0) and np.all(clen % SLICED_PAD == 0) and h["n_slots"] % 16 == 0
    assert h["pad_slot"] % SLICED_PAD == 0 and h["pad_slot"] + SLICED_PAD <= h["n_slots"]
    assert np.all(h["sub_chunk"] % 16 == 0) and chunks.shape[0] % 16 == 0
    # the run flags, checked by doing what a wave of csrc/spmm_sliced.hip does with them on one number per chunk:
    # two DPP steps inside groups of 4 quads, carries from group to group, the last quad of a run stores / adds
    flags = chunks[:, 1].astype(np.int64) & 0xFFFFFFFF
    in_row, carry, last, whole = (flags >> 24) & 3, (flags >> 26) & 1, (flags >> 27) & 1, (flags >> 28) & 1
    value = rng.integers(1, 1000, chunks.shape[0]).astype(np.int64) * real
    acc, plain_rows = np.zeros(n, dtype=np.int64), []
    group_of_chunk = np.searchsorted(h["sub_chunk"], np.arange(chunks.shape[0]), side="right") - 1
    for w0 in range(0, chunks.shape[0], 16):
        x = value[w0:w0 + 16].copy()
        x1 = x.copy()
        for j in range(16):
            if in_row[w0 + j] >= 1:
                assert j % 4 >= 1
                x1[j] = x[j] + x[j - 1]
        x2 = x1.copy()
        for j in range(16):
            if in_row[w0 + j] >= 2:
                assert j % 4 >= 2
                x2[j] = x1[j] + x1[j - 2]
        for r in range(1, 4):
            for j in range(4 * r, 4 * r + 4):
                if carry[w0 + j]:
                    x2[j] += x2[4 * r - 1]
        for j in range(16):
            if last[w0 + j]:
                assert real[w0 + j]
                acc[row[w0 + j]] += x2[j]
                if whole[w0 + j]:     # the kernel writes this row itself
                    plain_rows.append(row[w0 + j])
                else:                 # ... and sums this one in the workgroup's LDS accumulator `spill_idx`
                    gq = group_of_chunk[w0 + j]
                    assert spill_idx[w0 + j] < h["spill_ptr"][gq + 1] - h["spill_ptr"][gq] <= cap
                    assert h["spill_row"][h["spill_ptr"][gq] + spill_idx[w0 + j]] == row[w0 + j]
    want = np.zeros(n, dtype=np.int64)
    np.add.at(want, row[real], value[real])
    assert np.array_equal(acc, want), "the runs do not add up to the rows"
    assert len(set(plain_rows)) == len(plain_rows), "a row stored plainly twice"
    stores = np.bincount(row[last == 1], minlength=n)
    assert np.all(stores[np.array(plain_rows, dtype=np.int64)] == 1), "a plainly stored row has other contributions"
    # every row is written exactly once: by its one run, as a spill row of its workgroup, or as an empty row
    written = np.concatenate([np.array(plain_rows, dtype=np.int64), h["spill_row"], h["empty_row"]])
    assert np.array_equal(np.sort(written), np.arange(n))
    assert np.array_equal(h["empty_row"], np.nonzero(lens == 0)[0])
    for name in ("spill", "empty"):
        rows_of, ptr = h[name + "_row"], h[name + "_ptr"]
        assert ptr.size == n_groups + 1 and ptr[0] == 0 and ptr[-1] == rows_of.size
        for gq in range(n_groups):
            mine = rows_of[ptr[gq]:ptr[gq + 1]]
            assert np.all((mine >= h["sub_row"][gq]) & (mine < h["sub_row"][gq + 1]))
    start, row, clen = start[real], row[real], clen[real]
    covered = np.zeros(h["n_slots"], dtype=np.int32)
    slot_row = np.full(h["n_slots"], -1)
    for s, r, c in zip(start, row, clen):
        covered[s:s + c] += 1
        slot_row[s:s + c] = r
    assert np.all(covered[:h["pad_slot"]] == 1) and np.all(covered[h["pad_slot"]:] == 0)
    live = h["eid"] >= 0
    pad_col = n if "col_scale" in h else 0     # a factored graph's padding reads the all-zero source row
    assert live.sum() == nnz and np.all(h["val"][~live] == 0) and np.all(h["col16"][~live] == pad_col)
    back = np.argsort(eid)  # edge whose keep index is j
    edge_of_slot = back[h["eid"][live]]
    assert np.array_equal(np.sort(edge_of_slot), np.arange(nnz))
    assert np.array_equal(h["col16"][live], col[edge_of_slot].astype(np.uint16))
    assert np.array_equal(h["val"][live], val[edge_of_slot])
    assert np.array_equal(slot_row[live], np.repeat(np.arange(n), lens)[edge_of_slot])
    plain = sliced_graph_host(rowptr, col, val, None, n_groups, cap, lane_slots=lane_slots)
    assert np.array_equal(plain["eid"][live], np.arange(nnz))
    sub_row, sub_chunk = h["sub_row"], h["sub_chunk"]
    assert sub_row.size == n_groups * k + 1 and sub_row[0] == 0 and sub_row[-1] == n
    assert sub_chunk[0] == 0 and sub_chunk[-1] == h["n_chunks"] == chunks.shape[0]
    assert np.all(np.diff(sub_row) >= 0)
    all_rows = chunks[:, 1] & 0xFFFF
    for i in range(n_groups * k):
        rows_of = all_rows[sub_chunk[i]:sub_chunk[i + 1]]
        assert np.all((rows_of >= sub_row[i]) & (rows_of < sub_row[i + 1]))
    if heavy:   # the heavy row alone is several windows: a spill row, and so are the rows the window boundaries cut
        assert sliced_graph_host(rowptr, col, val, None, 1, 1, lane_slots=lane_slots) is None
    empty = sliced_graph_host(np.zeros(11, dtype=np.int64), col[:0], val[:0], None, 8, 16)
    assert empty["n_chunks"] == 0 and empty["pad_slot"] == 0 and empty["n_slots"] == 16 and empty["sub_row"][-1] == 10
    assert np.all(empty["sub_chunk"] == 0)


def test_factor_edge_values():
    """Rank-one form of a degree-normalised adjacency's values (column-only edge stream of the sliced SpMM)."""
    import scipy.sparse as sp

    from beta_recsys_amd.lightgcn import factor_edge_values, sliced_graph_host
    from oracle import lightgcn_numpy as olg

    rng = np.random.default_rng(5)
    U, I = 300, 170
    adj = olg.build_norm_adj(U, I, rng.integers(0, U, 4000), rng.integers(0, I, 4000))  # D^-1 (A + I)
    r, c = factor_edge_values(adj.indptr, adj.indices, adj.data)
    row = np.repeat(np.arange(U + I), np.diff(adj.indptr))
    assert np.allclose(r[row] * c[adj.indices], adj.data, rtol=3e-6, atol=0)
    assert np.allclose(c, c[0]) and r.dtype == np.float32  # row-normalised: the column factor is constant
    at = adj.T.tocsr()
    at.sort_indices()
    rt, ct = factor_edge_values(at.indptr, at.indices, at.data)
    rowt = np.repeat(np.arange(U + I), np.diff(at.indptr))
    assert np.allclose(rt[rowt] * ct[at.indices], at.data, rtol=3e-6, atol=0)
    # symmetric normalisation, several components, empty rows
    n = 400
    a = sp.random(n, n, density=0.01, random_state=3, format="csr")
    a.data[:] = 1.0
    a = ((a + a.T) > 0).astype(np.float64).tolil()
    a[50:60, :] = 0
    a[:, 50:60] = 0
    a = a.tocsr()
    a.eliminate_zeros()
    d = np.asarray(a.sum(1)).ravel()
    dinv = np.where(d > 0, 1 / np.sqrt(np.maximum(d, 1)), 0)
    sym = (sp.diags(dinv) @ a @ sp.diags(dinv)).tocsr().astype(np.float32)
    sym.sort_indices()
    rs, cs = factor_edge_values(sym.indptr, sym.indices, sym.data)
    rows = np.repeat(np.arange(n), np.diff(sym.indptr))
    assert np.allclose(rs[rows] * cs[sym.indices], sym.data, rtol=3e-6, atol=0)
    # arbitrary values do not factor; neither does a graph with an explicit zero
    rnd = sym.copy()
    rnd.data = rng.random(rnd.nnz).astype(np.float32) + 0.5
    assert factor_edge_values(rnd.indptr, rnd.indices, rnd.data) is None
    z = sym.copy()
    z.data[3] = 0
    assert factor_edge_values(z.indptr, z.indices, z.data) is None
    assert factor_edge_values(np.zeros(5, np.int64), np.zeros(0), np.zeros(0))[0].shape == (4,)
    # the host builder carries the factors and points padding slots at the zero row n
    h = sliced_graph_host(adj.indptr, adj.indices, adj.data, None, 8, 128)
    assert "row_scale" in h and np.all(h["col16"][h["eid"] < 0] == U + I)
    g = sliced_graph_host(adj.indptr, adj.indices, adj.data, None, 8, 128, factor=False)
    assert "row_scale" not in g and np.all(g["col16"][g["eid"] < 0] == 0)
    assert "row_scale" not in sliced_graph_host(rnd.indptr, rnd.indices, rnd.data, None, 8, 128)


def test_spread_bank_conflicts_only_reorders_inside_lane_segments():
    """The LDS bank-conflict permutation of a sliced graph: every lane segment (lane_slots slots) keeps its multiset
    of (column, value, keep index) -- sums are unchanged -- and the simulated conflict ways go down."""
    from beta_recsys_amd.lightgcn import sliced_graph_host, spread_bank_conflicts
    from oracle import lightgcn_numpy as olg

    rng = np.random.default_rng(9)
    U, I = 1500, 900
    adj = olg.build_norm_adj(U, I, rng.integers(0, U, 120_000), rng.integers(0, I, 120_000))
    for factor, S in ((True, None), (False, 16), (True, 16), (True, 24), (True, 48)):
        h = sliced_graph_host(adj.indptr, adj.indices, adj.data, None, 16, 256, factor=factor, lane_slots=S)
        S = h["lane_slots"]
        ref = {k: h[k].copy() for k in ("col16", "val", "eid")}
        before, after = spread_bank_conflicts(h, 16)
        assert after < 0.8 * before and after >= 1.0, (before, after)
        rows_end = h["pad_slot"]     # whole lane segments up to here, then the padding tail (left alone)
        assert all(np.array_equal(h[k][rows_end:], ref[k][rows_end:]) for k in ref)
        key_new = np.stack([h["col16"].astype(np.int64), h["eid"].astype(np.int64)], 1)[:rows_end].reshape(-1, S, 2)
        key_old = np.stack([ref["col16"].astype(np.int64), ref["eid"].astype(np.int64)], 1)[:rows_end].reshape(-1, S, 2)
        order_new = np.lexsort((key_new[:, :, 1], key_new[:, :, 0]), axis=1)
        order_old = np.lexsort((key_old[:, :, 1], key_old[:, :, 0]), axis=1)
        assert np.array_equal(np.take_along_axis(key_new, order_new[:, :, None], 1),
                              np.take_along_axis(key_old, order_old[:, :, None], 1))
        # values travel with their slots
        by_eid_new = dict(zip(h["eid"][h["eid"] >= 0].tolist(), h["val"][h["eid"] >= 0].tolist()))
        by_eid_old = dict(zip(ref["eid"][ref["eid"] >= 0].tolist(), ref["val"][ref["eid"] >= 0].tolist()))
        assert by_eid_new == by_eid_old
        assert not np.array_equal(h["col16"], ref["col16"])


def test_lazy_scalars_table_has_its_converged_tail_and_refuses_slow_betas():
    """beyond the tabulated steps a lazy Adam replay reads the table's LAST entry -- created with the
    converged pair (lr, 1) -- and betas whose bias corrections still move there are refused at set-up."""
    from beta_recsys_amd import _lib
    from beta_recsys_amd.torch_engine import HipOptimizer

    adam = HipOptimizer("adam", 0.05)
    assert _lib.lazy_betas_converge(adam) and _lib.lazy_betas_converge(HipOptimizer("rmsprop", 0.01))
    t = _lib.lazy_scalars_table(adam, "cpu")
    assert tuple(t.shape) == (1 << 16, 2) and t[:-1].abs().sum() == 0
    assert t[-1, 0].item() == np.float32(0.05) and t[-1, 1].item() == 1.0
    adam.beta2 = 0.9999      # 0.9999 ** 65535 = 1.4e-3: the corrections have not converged inside the table
    assert not _lib.lazy_betas_converge(adam)
    with pytest.raises(ValueError, match="converged"):
        _lib.lazy_scalars_table(adam, "cpu")
    assert _lib.lazy_scalars_table(HipOptimizer("rmsprop", 0.01), "cpu").abs().sum() == 0


def test_rccl_watchdog_wait_is_bounded_and_reports_asynchronous_errors():
    """_rccl.Communicator.wait: the host never blocks for ever on exchanges the C drivers posted --
    it polls the stream, raises on an asynchronous communicator error, and after the timeout aborts the communicator."""
    import ctypes

    from beta_recsys_amd import _rccl

    class FakeLib:
        def __init__(self, code):
            self.code, self.aborted = code, 0
            self.ncclAllReduce = ctypes.CFUNCTYPE(ctypes.c_int)(lambda: 0)   # (the constructor takes its address)

        def ncclCommGetAsyncError(self, comm, out):
            ctypes.cast(out, ctypes.POINTER(ctypes.c_int))[0] = self.code
            return 0

        def ncclCommAbort(self, comm):
            self.aborted += 1
            return 0

    class Stream:
        def __init__(self, done_after):
            self.calls, self.done_after = 0, done_after

        def query(self):
            self.calls += 1
            return self.calls > self.done_after

    lib = FakeLib(0)
    comm = _rccl.Communicator(lib, ctypes.c_void_p(1), 2, 0)
    comm.wait(Stream(5), timeout_s=5.0)                       # completes: no error, nothing aborted
    assert lib.aborted == 0 and comm.comm
    with pytest.raises(RuntimeError, match="did not complete within"):
        comm.wait(Stream(10**9), timeout_s=0.05)              # a peer never posts: bounded, communicator aborted
    assert lib.aborted == 1 and comm.comm is None
    lib2 = FakeLib(5)                                          # an asynchronous error code from the communicator
    comm2 = _rccl.Communicator(lib2, ctypes.c_void_p(1), 2, 1)
    with pytest.raises(RuntimeError, match="asynchronous error 5"):
        comm2.wait(Stream(10**9), timeout_s=5.0)
    assert lib2.aborted == 1
    comm3 = _rccl.Communicator(FakeLib(7), ctypes.c_void_p(1), 2, 0)   # ncclInProgress is not an error
    comm3.wait(Stream(3), timeout_s=5.0)
