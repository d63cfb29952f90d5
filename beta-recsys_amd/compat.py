"""Route beta-recsys through the HIP engines without editing it.

``install()`` registers this package's mirrors under the reference's module names, so that
``from ..models.mf import MFEngine`` inside ``beta_rec/recommenders/matrix_factorization.py:8``
(and ``from beta_rec.models.torch_engine import ModelEngine``) resolve to them.  Call it before
importing ``beta_rec.recommenders``.
"""
import sys

# reference module name -> mirror module in this package
MIRRORS = {
    "beta_rec.models.torch_engine": "torch_engine",
    "beta_rec.models.mf": "mf",
    "beta_rec.models.ncf": "ncf",
    "beta_rec.models.gmf": "ncf",
    "beta_rec.models.mlp": "ncf",
    "beta_rec.models.lightgcn": "lightgcn",
    "beta_rec.models.ngcf": "ngcf",
    "beta_rec.models.pairwise_gmf": "pairwise_gmf",
    "beta_rec.models.triple2vec": "triple2vec",
}


def install(extra=None):
    """Register the mirrors in ``sys.modules``; returns the list of module names replaced."""
    import importlib

    pkg = __name__.rsplit(".", 1)[0]
    table = dict(MIRRORS)
    table.update(extra or {})
    done = []
    for ref_name, local in table.items():
        mod = importlib.import_module(f"{pkg}.{local}")
        sys.modules[ref_name] = mod
        parent_name, _, attr = ref_name.rpartition(".")
        parent = sys.modules.get(parent_name)
        if parent is not None:
            setattr(parent, attr, mod)
        done.append(ref_name)
    return done


def uninstall():
    """Remove the registrations made by :func:`install` (the reference's own modules load again)."""
    for ref_name in MIRRORS:
        mod = sys.modules.get(ref_name)
        if mod is not None and mod.__name__.startswith(__name__.rsplit(".", 1)[0]):
            del sys.modules[ref_name]


def install_eval():
    """Route ``beta_rec.core.eval_engine.evaluate`` (eval_engine.py:49-87; called by
    ``train_eval_worker`` :91-141 and ``test_eval_worker`` :145-170) through the HIP ranking-metric
    kernel.  Calls that ask for a non-ranking metric (rmse, mae, rsquared) keep going to the
    reference's own function — those are not part of this path."""
    import importlib

    from . import eval as hip_eval

    ee = importlib.import_module("beta_rec.core.eval_engine")
    if getattr(ee.evaluate, "_hiprec", False):
        return ee.evaluate
    reference_evaluate = ee.evaluate

    def evaluate(data_df, predictions, metrics, k_li):
        if all(m in hip_eval.RANK_METRICS for m in metrics):
            return hip_eval.evaluate(data_df, predictions, metrics, k_li)
        return reference_evaluate(data_df, predictions, metrics, k_li)

    evaluate._hiprec = True
    evaluate._reference = reference_evaluate
    ee.evaluate = evaluate
    return evaluate


def uninstall_eval():
    ee = sys.modules.get("beta_rec.core.eval_engine")
    if ee is not None and getattr(ee.evaluate, "_hiprec", False):
        ee.evaluate = ee.evaluate._reference


_DATA_METHODS = ("instance_bpr_loader", "instance_bce_loader", "instance_mul_neg_loader")


def install_data():
    """Route ``BaseData.instance_{bpr,bce,mul_neg}_loader`` (data/base_data.py:182-288) through the
    device-side sampler + batchers of :mod:`beta_recsys_amd.data`.  Same call signatures
    (``data.instance_bpr_loader(batch_size, device)`` ...); the returned loaders are device batchers
    that the engines of this package run resident."""
    import importlib

    from . import data as hip_data

    base = importlib.import_module("beta_rec.data.base_data").BaseData
    for name in _DATA_METHODS:
        current = getattr(base, name)
        if getattr(current, "_hiprec", False):
            continue
        hip_fn = getattr(hip_data, name)

        def method(self, *args, _fn=hip_fn, **kwargs):
            return _fn(self, *args, **kwargs)

        method._hiprec = True
        method._reference = current
        method.__name__ = name
        method.__doc__ = hip_fn.__doc__
        setattr(base, name, method)
    return base


def uninstall_data():
    mod = sys.modules.get("beta_rec.data.base_data")
    if mod is None:
        return
    for name in _DATA_METHODS:
        current = getattr(mod.BaseData, name, None)
        if getattr(current, "_hiprec", False):
            setattr(mod.BaseData, name, current._reference)
