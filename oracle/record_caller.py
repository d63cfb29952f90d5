"""Record what the REAL beta-recsys caller does to an engine (test infrastructure; runs in the build container only).

    python oracle/record_caller.py            # writes tests/golden/caller_transcript_{mf,ncf,lightgcn}.json

The drop-in boundary of this repository is the duck type of ``beta_rec.models.*Engine`` as the reference's own
callers use it: ``beta_rec.recommenders.MatrixFactorization.train`` (recommenders/matrix_factorization.py:51-110)
-> ``TrainEngine._train`` (core/train_engine.py:225-240) -> ``EvalEngine.train_eval`` (core/eval_engine.py)
and ``Recommender.load / predict`` (core/recommender.py:46-70).  This script runs exactly that code -- the real
recommender classes, TrainEngine, EvalEngine and BaseData of /root/reference, on the CPU, on a small synthetic
data set -- with a RECORDING engine installed under the reference's module name.  The recording engine is the
reference's own engine (so the run completes and every returned value is real) wrapped so that every attribute
read, method call, argument (type / dtype / shape) and return value the callers touch is written down, together
with the config keys the engine constructor reads.  ``tests/test_caller_transcript.py`` replays the transcript
against this package's engines on the GPU (and checks the host-side surface on the CPU).

Modules the reference imports but this image lacks (ray, munch, GPUtil, tensorboardX, py7zr, aiofiles, cpuinfo)
are replaced by inert stand-ins in ``sys.modules`` of THIS process: none of them is on the recorded path
(ray.tune is only used by ``tune_train``, GPUtil by the GPU picker which is bypassed with ``device: cpu``).
No reference source is copied: the transcript holds names, types, shapes and scalar values only.
"""
import importlib
import json
import os
import sys
import tempfile
import threading
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT_DIR = os.path.join(ROOT, "tests", "golden")


# ---- stand-ins for modules that are absent here -------------------------------------------------------
class Munch(dict):
    """munch.Munch: a dict whose keys are attributes too (``vars()`` of it is the mapping itself, which is what
    utils/common_util.py:54-67 ``update_args`` iterates)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.__dict__ = self


def munchify(x):
    if isinstance(x, dict):
        return Munch((k, munchify(v)) for k, v in x.items())
    if isinstance(x, (list, tuple)):
        return type(x)(munchify(v) for v in x)
    return x


class _Inert:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __getattr__(self, k):
        return _Inert()


def install_stand_ins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []

        def _missing(k):
            if k.startswith("__"):   # inspect / importlib probe dunder attributes of every module
                raise AttributeError(k)
            return _Inert()

        m.__getattr__ = _missing
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("munch", Munch=Munch, munchify=munchify)
    ray = mod("ray")
    ray.tune = mod("ray.tune")
    ray.utils = mod("ray.utils")
    mod("GPUtil", getAvailable=lambda **k: [])

    class SummaryWriter:
        def __init__(self, log_dir=None, **k):
            self.log_dir = log_dir

        def __getattr__(self, name):   # add_scalar / add_text / close ...: accepted and dropped
            return lambda *a, **k: None

    mod("tensorboardX", SummaryWriter=SummaryWriter)
    mod("cpuinfo", get_cpu_info=lambda: {"brand_raw": "stand-in"})   # utils/monitor.py:34-41 formats it
    for name in ("py7zr", "aiofiles"):
        mod(name)


# ---- describing values -----------------------------------------------------------------------------------
def describe(v, depth=0):
    import pandas as pd
    import torch

    if isinstance(v, torch.Tensor):
        return {"type": "torch.Tensor", "dtype": str(v.dtype), "shape": list(v.shape), "device": v.device.type}
    if isinstance(v, np.ndarray):
        return {"type": "numpy.ndarray", "dtype": str(v.dtype), "shape": list(v.shape)}
    if isinstance(v, pd.DataFrame):
        return {"type": "pandas.DataFrame", "columns": list(map(str, v.columns)), "rows": len(v)}
    if isinstance(v, torch.utils.data.DataLoader):
        ds = v.dataset
        return {"type": "torch.utils.data.DataLoader", "batch_size": v.batch_size, "dataset": type(ds).__name__,
                "dataset_fields": {k: describe(getattr(ds, k)) for k in vars(ds) if not k.startswith("_")},
                "shuffle": type(v.sampler).__name__ == "RandomSampler", "n_batches": len(v)}
    if isinstance(v, (bool, int, float, str)) or v is None:
        return {"type": type(v).__name__, "value": v}
    if isinstance(v, (list, tuple)) and depth < 2:
        return {"type": type(v).__name__, "items": [describe(x, depth + 1) for x in v[:8]], "len": len(v)}
    if isinstance(v, dict) and depth < 2:
        return {"type": type(v).__name__, "keys": sorted(map(str, v.keys()))[:40]}
    return {"type": type(v).__module__ + "." + type(v).__name__}


VALUES = {}   # config key -> description of the value the engine saw


class Transcript:
    def __init__(self):
        self.events = []
        self.lock = threading.Lock()
        self.main = threading.get_ident()

    def add(self, **ev):
        ev["thread"] = "main" if threading.get_ident() == self.main else "eval-worker"
        with self.lock:
            self.events.append(ev)


class RecordingDict(dict):
    """config / config['model'] / config['system'] as handed to the engine: records which keys are read."""

    def __init__(self, data, path, reads):
        super().__init__(data)
        self._path, self._reads = path, reads
        for k, v in list(self.items()):
            if isinstance(v, dict):
                dict.__setitem__(self, k, RecordingDict(v, f"{path}.{k}", reads))

    def _note(self, k, how):
        self._reads.setdefault(f"{self._path}.{k}", set()).add(how)
        if dict.__contains__(self, k):
            v = dict.__getitem__(self, k)
            VALUES[f"{self._path}.{k}"] = ({"type": "dict"} if isinstance(v, dict) else
                                          {"type": "list", "value": list(v)} if isinstance(v, (list, tuple)) and
                                          all(isinstance(x, (int, float, str)) for x in v) else describe(v))

    def __getitem__(self, k):
        self._note(k, "getitem" if dict.__contains__(self, k) else "getitem-missing")
        return super().__getitem__(k)

    def __contains__(self, k):
        self._note(k, "in")
        return super().__contains__(k)

    def get(self, k, default=None):
        self._note(k, "get")
        return super().get(k, default)

    def __getattr__(self, k):  # Munch-style access
        if k.startswith("_"):
            raise AttributeError(k)
        return self[k]


class ModelProxy:
    """engine.model as the callers see it: records attribute reads and calls, forwards to the real model."""

    def __init__(self, real, tr):
        object.__setattr__(self, "_real", real)
        object.__setattr__(self, "_tr", tr)

    def __getattr__(self, name):
        val = getattr(self._real, name)
        if callable(val) and not isinstance(val, type):
            def call(*a, **k):
                out = val(*a, **k)
                self._tr.add(obj="engine.model", call=name, args=[describe(x) for x in a],
                             kwargs={kk: describe(vv) for kk, vv in k.items()}, returns=describe(out))
                return out
            return call
        self._tr.add(obj="engine.model", read=name, value=describe(val))
        return val

    def __setattr__(self, name, value):
        setattr(self._real, name, value)


def make_recording_engine(real_cls, tr, config_reads, exposed=("model",)):
    """Subclass of the reference's engine: same behaviour, every external call / attribute read written down."""
    recorded_methods = ("train_an_epoch", "train_single_batch", "save_checkpoint", "resume_checkpoint")

    class Recording(real_cls):
        def __init__(self, config):
            cfg = RecordingDict(dict(config), "config", config_reads)
            tr.add(obj="engine", call="__init__", args=[{"type": type(config).__name__}])
            object.__setattr__(self, "_rec_inside", 1)
            try:
                super().__init__(cfg)
            finally:
                object.__setattr__(self, "_rec_inside", 0)

        def __getattribute__(self, name):
            val = object.__getattribute__(self, name)
            if name.startswith("_") or object.__getattribute__(self, "_rec_inside"):
                return val
            if name in recorded_methods:
                def call(*a, **k):
                    object.__setattr__(self, "_rec_inside", 1)
                    try:
                        out = val(*a, **k)
                    finally:
                        object.__setattr__(self, "_rec_inside", 0)
                    tr.add(obj="engine", call=name, args=[describe(x) for x in a],
                           kwargs={kk: describe(vv) for kk, vv in k.items()}, returns=describe(out))
                    return out
                return call
            if name in exposed:
                tr.add(obj="engine", read=name, value=describe(val))
                return ModelProxy(val, tr)
            tr.add(obj="engine", read=name, value=describe(val))
            return val

    Recording.__name__ = real_cls.__name__
    return Recording


# ---- a small synthetic split in the reference's own frame layout ----------------------------------------------
def synthetic_split(n_users=60, n_items=150, per_user=12, n_neg=20, seed=0):
    import pandas as pd

    rng = np.random.default_rng(seed)
    rows, valid, test = [], [], []
    for u in range(n_users):
        items = rng.permutation(n_items)
        pos = items[:per_user]
        for t, i in enumerate(pos[:-2]):
            rows.append((u + 100, int(i) + 1000, float(rng.integers(1, 6)), t))
        for held, frame in ((pos[-2], valid), (pos[-1], test)):
            frame.append((u + 100, int(held) + 1000, 1.0, 10_000))
            for i in items[per_user:per_user + n_neg]:
                frame.append((u + 100, int(i) + 1000, 0.0, 10_000))
    cols = ["col_user", "col_item", "col_rating", "col_timestamp"]
    mk = lambda r: pd.DataFrame(r, columns=cols)  # noqa: E731
    return mk(rows), [mk(valid)], [mk(test)]


def run_recommender(kind, tmp):
    """Run the reference's recommender `kind` end to end on the CPU with a recording engine; return the transcript."""
    tr, config_reads = Transcript(), {}
    VALUES.clear()
    module_name, engine_names, rec_mod, rec_cls, config_file = {
        "mf": ("beta_rec.models.mf", ["MFEngine"], "beta_rec.recommenders.matrix_factorization", "MatrixFactorization",
               "mf_default.json"),
        "ncf": ("beta_rec.models.ncf", ["NeuMFEngine"], "beta_rec.recommenders.ncf", "NeuCF", "ncf_default.json"),
        "lightgcn": ("beta_rec.models.lightgcn", ["LightGCNEngine"], "beta_rec.recommenders.lightgcn", "LightGCN",
                     "lightgcn_default.json"),
    }[kind]
    # the recommender module binds the engine classes by name (`from ..models.mf import MFEngine`): the recording
    # subclasses go into ITS namespace (inside the model module the reference's own `super(MFEngine, self)` must
    # keep resolving to the real class)
    rec = importlib.import_module(rec_mod)
    for name in dir(rec):
        obj = getattr(rec, name)
        if isinstance(obj, type) and name.endswith("Engine") and obj.__module__.startswith("beta_rec.models."):
            setattr(rec, name, make_recording_engine(obj, tr, config_reads))
    from beta_rec.data.base_data import BaseData

    args = {"config_file": os.path.join(REF, "configs", config_file), "root_dir": tmp, "device": "cpu",
            "max_epoch": 2, "batch_size": 256, "emb_dim": 16}
    if kind == "ncf":
        args.update(model="ncf_end")   # the end-to-end NeuMF mode (no pre-trained GMF / MLP checkpoints needed)
    recommender = getattr(rec, rec_cls)(munchify(args))
    data = BaseData(synthetic_split())
    notes = []
    if kind == "ncf":
        # Reference defect (quirk Q13): recommenders/ncf.py:78-81 builds its loader with instance_bpr_loader, whose
        # (user, pos_item, neg_item) batches NeuMFEngine.train_an_epoch (models/ncf.py) unpacks as (user, item,
        # rating): BCELoss then raises on targets outside [0, 1] and NeuCF.train never completes.  The working
        # caller, examples/train_ncf.py:120-131, feeds (user, item, rating) batches with sampled negatives, which is
        # what instance_bce_loader (data/base_data.py:182-216) builds; the rest of NeuCF.train runs unchanged.
        data.instance_bpr_loader = lambda batch_size, device: data.instance_bce_loader(batch_size, device, 4)
        notes.append("NeuCF.train's instance_bpr_loader replaced by instance_bce_loader(num_negative=4): the "
                     "recommender's own loader makes BCELoss raise (reference defect, see oracle/record_caller.py)")
    tr.add(obj="caller", note="recommender.train(data)", data={"n_users": data.n_users, "n_items": data.n_items})
    result = recommender.train(data)
    while recommender.eval_engine.n_worker > 0:
        time.sleep(0.2)
    tr.add(obj="caller", note="train returned", returns=describe(result))
    model_dir = result["model_save_dir"] if isinstance(result, dict) and "model_save_dir" in result else None
    if model_dir and os.path.exists(model_dir):
        recommender.load(model_dir)       # core/recommender.py:46-56 -> engine.resume_checkpoint
    test_df = data.test[0]
    out = recommender.predict(test_df)   # core/recommender.py:58-70 -> eval_engine.predict -> model.predict
    tr.add(obj="caller", note="recommender.predict(test_df)", returns=describe(out))
    reads = {k: sorted(v) for k, v in sorted(config_reads.items())}
    # distinct events (what is called / read, with which argument and return types) in first-seen order, and how
    # often; the order of the first epoch is kept verbatim
    distinct, counts = [], {}
    for ev in tr.events:
        key = json.dumps({k: v for k, v in ev.items() if k != "thread"}, sort_keys=True, default=str)
        key = key.replace(tmp, "<root_dir>")
        if key not in counts:
            distinct.append(key)
        counts[key] = counts.get(key, 0) + 1
    return {"recommender": f"{rec_mod}.{rec_cls}", "engine_module": module_name, "config_file": config_file,
            "args": {k: (v.replace(REF, "<reference>").replace(tmp, "<root_dir>") if isinstance(v, str) else v)
                     for k, v in args.items()},
            "config_keys_read_by_engine": reads, "config_values_seen_by_engine": json.loads(json.dumps(dict(sorted(VALUES.items())), default=str).replace(tmp, "<root_dir>")),
            "notes": notes,
            "distinct_events": [dict(json.loads(k), count=counts[k]) for k in distinct],
            "sequence_head": [json.loads(json.dumps(ev, default=str).replace(tmp, "<root_dir>")) for ev in tr.events[:24]]}


def main():
    sys.path.insert(0, REF)
    install_stand_ins()
    for alias, t in (("int", int), ("float", float), ("bool", bool)):   # the reference predates numpy 1.24
        if not hasattr(np, alias):
            setattr(np, alias, t)
    import torch

    torch.manual_seed(0)
    kinds = sys.argv[1:] or ["mf", "ncf", "lightgcn"]
    for kind in kinds:
        with tempfile.TemporaryDirectory() as tmp:
            cwd = os.getcwd()
            os.chdir(tmp)
            try:
                transcript = run_recommender(kind, tmp)
            except BaseException:
                import traceback

                with open("/tmp/record_caller_error.log", "w") as ef:   # the reference's logger swallows sys.stderr
                    traceback.print_exc(file=ef)
                print("FAILED: see /tmp/record_caller_error.log", file=sys.__stdout__, flush=True)
                os._exit(1)                                # (its Monitor thread would keep the process alive)
            finally:
                os.chdir(cwd)
                sys.stdout, sys.stderr = sys.__stdout__, sys.__stderr__   # utils/logger.py redirected them into tmp
        path = os.path.join(OUT_DIR, f"caller_transcript_{kind}.json")
        with open(path, "w") as f:
            json.dump(transcript, f, indent=1, default=str)
        print(f"wrote {path}: {len(transcript['distinct_events'])} distinct events")


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        import traceback

        with open("/tmp/record_caller_error.log", "w") as ef:
            traceback.print_exc(file=ef)
        os._exit(1)
    os._exit(0)   # the reference's Monitor / eval threads would keep the interpreter alive
