"""hiprec — MI355X-native embedding-table training hot path for beta-recsys models.

Host-side mirrors of the reference's engine modules (``beta_rec.models.{torch_engine,mf,...}``)
over the C-ABI HIP library ``libhiprec.so`` (``csrc/``, declared in ``include/hiprec.h``).
Import as ``beta_recsys_amd`` (see the shim package of that name at the repository root).
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401
from .torch_engine import HipOptimizer, ModelEngine  # noqa: F401
from .mf import MF, DeviceTripleBatcher, MFEngine, gather_rows  # noqa: F401
from .ncf import GMF, MLP, GMFEngine, MLPEngine, NeuMF, NeuMFEngine  # noqa: F401
from .lightgcn import LightGCN, LightGCNEngine  # noqa: F401
from .ngcf import NGCF, NGCFEngine  # noqa: F401
from .pairwise_gmf import PairwiseGMF, PairwiseGMFEngine  # noqa: F401
from .triple2vec import Triple2vec, Triple2vecEngine  # noqa: F401
from . import eval  # noqa: F401,A004  (evaluate / predict / rank_metrics)
from . import data  # noqa: F401  (device-side loaders / negative sampling)
