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
