"""Route beta-recsys through the HIP engines without editing it.

``install()`` registers this package's mirrors under the reference's module names, so that
``from ..models.mf import MFEngine`` inside ``beta_rec/recommenders/matrix_factorization.py:8``
resolves to them.  Call it before
importing ``beta_rec.recommenders``.
"""
import sys

# reference module name -> mirror module in this package.  ``beta_rec.models.torch_engine`` is NOT in
# this table: about fifteen reference engines this package does not mirror (vbcar, narm, sasrec, sgl,
# ultragcn, vaecf, cmn ...) import ``ModelEngine`` from it and need the torch optimizer it builds
# (``self.optimizer.step()``); the mirrors import their own base class relatively.
MIRRORS = {
    "beta_rec.models.mf": "mf",
    "beta_rec.models.ncf": "ncf",
    "beta_rec.models.gmf": "ncf",
    "beta_rec.models.mlp": "ncf",
    "beta_rec.models.lightgcn": "lightgcn",
    "beta_rec.models.ngcf": "ngcf",
    "beta_rec.models.pairwise_gmf": "pairwise_gmf",
    "beta_rec.models.triple2vec": "triple2vec",
}

_MISSING = object()
_saved = {}  # ref_name -> (previous sys.modules entry, previous parent attribute)


def install(extra=None):
    """Register the mirrors in ``sys.modules``; returns the list of module names replaced.

    ``extra`` adds entries, e.g. ``{"beta_rec.models.torch_engine": "torch_engine"}`` for a caller
    that wants the HIP ``ModelEngine`` base under the reference's name as well (its
    ``set_optimizer`` falls back to ``torch.optim`` for models without a flat parameter buffer, so
    un-mirrored engines keep training)."""
    import importlib

    pkg = __name__.rsplit(".", 1)[0]
    table = dict(MIRRORS)
    table.update(extra or {})
    done = []
    for ref_name, local in table.items():
        mod = importlib.import_module(f"{pkg}.{local}")
        parent_name, _, attr = ref_name.rpartition(".")
        parent = sys.modules.get(parent_name)
        if ref_name not in _saved:
            _saved[ref_name] = (sys.modules.get(ref_name, _MISSING),
                                _MISSING if parent is None else parent.__dict__.get(attr, _MISSING))
        sys.modules[ref_name] = mod
        if parent is not None:
            setattr(parent, attr, mod)
        done.append(ref_name)
    return done


def uninstall():
    """Undo :func:`install`: ``sys.modules`` entries and the parent packages' attributes go back to
    what they were (the reference's own modules load again)."""
    for ref_name, (prev_mod, prev_attr) in list(_saved.items()):
        if prev_mod is _MISSING:
            sys.modules.pop(ref_name, None)
        else:
            sys.modules[ref_name] = prev_mod
        parent_name, _, attr = ref_name.rpartition(".")
        parent = sys.modules.get(parent_name)
        if parent is not None:
            if prev_attr is not _MISSING:
                setattr(parent, attr, prev_attr)
            elif prev_mod is not _MISSING:
                setattr(parent, attr, prev_mod)
            elif attr in parent.__dict__:
                delattr(parent, attr)
        del _saved[ref_name]


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
