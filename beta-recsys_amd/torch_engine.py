"""Host-side mirror of beta_rec/models/torch_engine.py:6-121 (``ModelEngine``).

Same constructor contract, attributes (``model optimizer device writer config``) and methods
(``save_checkpoint resume_checkpoint bpr_loss bce_loss``) as the reference base class, so
``beta_rec.core.train_engine.TrainEngine._train`` (core/train_engine.py:225-240) and
``Recommender.load`` (core/recommender.py:46-56) keep working unchanged.  The optimizer object is a
thin descriptor: the arithmetic of ``torch.optim.{SGD,Adam,RMSprop}.step`` runs in
``csrc/optim.hip`` on flat state buffers owned here.
"""
import torch
import torch.nn.functional as F

from . import _lib


class _NullWriter:
    """Stand-in for tensorboardX.SummaryWriter when tensorboardX is not installed."""

    def __init__(self, log_dir=None):
        self.log_dir = log_dir
        self.scalars = []

    def add_scalar(self, tag, value, step=None):
        self.scalars.append((tag, value, step))

    def close(self):
        pass


def make_writer(log_dir):
    """SummaryWriter(log_dir=...) as torch_engine.py:19-21, or a recording no-op."""
    try:
        from tensorboardX import SummaryWriter  # type: ignore

        return SummaryWriter(log_dir=log_dir)
    except Exception:  # tensorboardX is absent in this image
        return _NullWriter(log_dir)


class HipOptimizer:
    """Descriptor + state of the optimizer torch_engine.py:23-39 would have built.

    Only ``lr`` is configurable in the reference; everything else is the torch default:
    Adam betas (0.9, 0.999) eps 1e-8; RMSprop alpha 0.99 eps 1e-8; SGD momentum 0.
    State lives in flat fp32 buffers laid out like the model's flat parameter buffer.
    """

    DEFAULTS = {
        "sgd": dict(beta1=0.0, beta2=0.0, eps=0.0),
        "adam": dict(beta1=0.9, beta2=0.999, eps=1e-8),
        "rmsprop": dict(beta1=0.0, beta2=0.99, eps=1e-8),  # beta2 plays alpha
    }

    def __init__(self, name, lr):
        if name not in _lib.OPT_KINDS:
            raise ValueError(
                f"Unsupported optimizer {name!r}: the engine supports 'sgd', 'adam', 'rmsprop' "
                "(beta_rec/models/torch_engine.py:23-39)"
            )
        self.name = name
        self.kind = _lib.OPT_KINDS[name]
        self.lr = float(lr)
        d = self.DEFAULTS[name]
        self.beta1, self.beta2, self.eps = d["beta1"], d["beta2"], d["eps"]
        self.exp_avg = None      # Adam m           (flat, zeros)
        self.exp_avg_sq = None   # Adam v / RMSprop square_avg
        self.defaults = {"lr": self.lr}
        if name == "adam":
            self.defaults.update(betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False)
        elif name == "rmsprop":
            self.defaults.update(alpha=0.99, eps=1e-8, weight_decay=0, momentum=0, centered=False)
        else:
            self.defaults.update(momentum=0, dampening=0, weight_decay=0, nesterov=False)
        self.param_groups = [dict(self.defaults)]

    def allocate_state(self, flat_like):
        """Zero state buffers matching the flat parameter buffer."""
        if self.name == "adam":
            self.exp_avg = torch.zeros_like(flat_like)
            self.exp_avg_sq = torch.zeros_like(flat_like)
        elif self.name == "rmsprop":
            self.exp_avg_sq = torch.zeros_like(flat_like)

    def zero_grad(self):
        """No-op: the HIP optimizer kernels clear the gradient buffer as they consume it."""

    def state_dict(self):
        """Flat moment buffers + hyper-parameters (the reference never persists optimizer state,
        torch_engine.py:70-73; offered for resume)."""
        return {"name": self.name, "lr": self.lr,
                "exp_avg": None if self.exp_avg is None else self.exp_avg.clone(),
                "exp_avg_sq": None if self.exp_avg_sq is None else self.exp_avg_sq.clone()}

    def load_state_dict(self, sd):
        if sd["name"] != self.name:
            raise ValueError(f"optimizer state is for {sd['name']!r}, engine uses {self.name!r}")
        self.lr = float(sd["lr"])
        for attr in ("exp_avg", "exp_avg_sq"):
            if sd[attr] is not None and getattr(self, attr) is not None:
                getattr(self, attr).copy_(sd[attr])

    def __repr__(self):
        return f"HipOptimizer({self.name}, lr={self.lr})"


class ModelEngine(object):
    """Meta engine; subclasses must set ``self.model`` before calling ``__init__``."""

    def __init__(self, config):
        """Mirror of torch_engine.py:12-21: device, optimizer, model.to(device), writer."""
        self.config = config
        self.set_device()
        self.set_optimizer()
        self.model.to(self.device)
        print(self.model)
        self.writer = make_writer(config["system"]["run_dir"])

    def set_optimizer(self):
        """torch_engine.py:23-39; unknown names raise ValueError early instead of AttributeError later.

        Flat-buffer models (everything this package mirrors) get the HIP optimizer descriptor.  A
        model WITHOUT a flat buffer -- one of the reference's own engines subclassing this base
        through ``compat.install(extra={"beta_rec.models.torch_engine": ...})`` -- gets exactly the
        torch optimizer the reference builds, so its ``optimizer.step()`` keeps working."""
        name, lr = self.config["model"]["optimizer"], self.config["model"]["lr"]
        if hasattr(self.model, "flat"):
            self.optimizer = HipOptimizer(name, lr)
            return
        params = self.model.parameters()
        if name == "sgd":
            self.optimizer = torch.optim.SGD(params, lr=lr)
        elif name == "adam":
            self.optimizer = torch.optim.Adam(params, lr=lr)
        elif name == "rmsprop":
            self.optimizer = torch.optim.RMSprop(params, lr=lr)
        else:
            raise ValueError(f"Unsupported optimizer {name!r} (beta_rec/models/torch_engine.py:23-39)")

    def set_device(self):
        """torch_engine.py:41-45."""
        self.device = torch.device(self.config["model"]["device_str"])
        self.model.device = self.device
        print("Setting device for torch_engine", self.device)

    def require_hip(self):
        """The compute path is HIP only; refuse to run anywhere else."""
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError(
                f"hiprec engines run on an MI355X through libhiprec.so; device {self.device} has no "
                "HIP path and there is deliberately no CPU fallback"
            )
        return _lib.load()

    def train_single_batch(self, batch_data, ratings=None):
        """torch_engine.py:47-56 is a generic template every reference engine overrides."""
        raise NotImplementedError("engines implement train_single_batch")

    def train_an_epoch(self, train_loader, epoch_id):
        """torch_engine.py:58-68: likewise overridden by every engine."""
        raise NotImplementedError("engines implement train_an_epoch")

    OPT_STATE_SUFFIX = ".opt"          # the optional second file of a checkpoint: optimizer clock + moments
    OPT_STATE_FORMAT = "hiprec-optimizer-state-1"

    def save_checkpoint(self, model_dir, optimizer_state=False):
        """torch_engine.py:70-73: torch.save(model.state_dict(), path) -- that file is all the reference writes and
        stays byte-compatible with it.  ``optimizer_state=True`` (SURVEY 8f-3's optional half) also writes
        ``model_dir + ".opt"``: the optimizer clock exactly as the device holds it, the moments and, for the lazy
        Adam / RMSprop form, the per-row stamps and the per-step scalars table after a flush -- what
        :meth:`resume_checkpoint` needs to continue the run bit for bit."""
        assert hasattr(self, "model"), "Please specify the exact model !"
        if hasattr(self, "flush_lazy"):
            self.flush_lazy()           # a lagging row's weights on disk must be the dense steps' weights
        torch.save(self.model.state_dict(), model_dir)
        if optimizer_state:
            torch.save(self.optimizer_checkpoint(), model_dir + self.OPT_STATE_SUFFIX)

    def resume_checkpoint(self, model_dir, model=None, optimizer_state=False):
        """torch_engine.py:76-90.  ``optimizer_state=True`` also restores ``model_dir + ".opt"`` (written by
        ``save_checkpoint(..., optimizer_state=True)``) into this engine."""
        assert hasattr(self, "model"), "Please specify the exact model !"
        print("loading model from:", model_dir)
        state_dict = torch.load(model_dir, map_location=self.device)
        target = self.model if model is None else model
        own = target is self.model
        if optimizer_state and not own:
            raise ValueError("the optimizer state belongs to the engine's own model")
        if own and hasattr(self, "flush_lazy"):
            # a lagging row's pending zero-gradient steps belong to the OLD weights and moments: replay them before
            # the weights are replaced, never onto the restored ones
            self.flush_lazy()
        target.load_state_dict(state_dict)
        target.to(self.device)
        if optimizer_state:
            self.load_optimizer_checkpoint(torch.load(model_dir + self.OPT_STATE_SUFFIX, map_location="cpu"))
        elif own and getattr(self, "_lazy", None) is not None:
            # weights alone were restored: they are what they are as of the engine's clock, nothing is owed to them
            self._lazy_mark_current()
            self._lazy["dirty"] = False
        return target

    def optimizer_checkpoint(self):
        """Everything beyond the weights that the next step depends on, as CPU tensors: the raw ``hiprec_stats`` block
        (step count and the running beta powers as the device's doubles -- recomputing them with pow() differs in the
        last bits), the flat moment buffers, and the lazy form's stamps / scalars table (rows are flushed first: every
        touched row is current as of the clock)."""
        if not isinstance(getattr(self, "optimizer", None), HipOptimizer):
            raise NotImplementedError("optimizer checkpoints exist for the flat-buffer engines")
        self._setup()
        if hasattr(self, "flush_lazy"):
            self.flush_lazy()
        opt = self.optimizer
        cpu = lambda t: None if t is None else t.detach().to("cpu", copy=True)   # noqa: E731
        lz = getattr(self, "_lazy", None)
        return {"format": self.OPT_STATE_FORMAT, "optimizer": opt.name, "lr": opt.lr,
                "n_params": int(self.model.flat.numel()),
                "stats": cpu(self._stats.view(torch.uint8)),
                "exp_avg": cpu(opt.exp_avg), "exp_avg_sq": cpu(opt.exp_avg_sq),
                "lazy": None if lz is None else {k: cpu(lz[k]) for k in ("stamp_u", "stamp_i", "scalars")}}

    def load_optimizer_checkpoint(self, payload):
        """Counterpart of :meth:`optimizer_checkpoint`; the weights are restored by ``load_state_dict``."""
        if payload.get("format") != self.OPT_STATE_FORMAT:
            raise ValueError(f"not a {self.OPT_STATE_FORMAT} file")
        self._setup()
        opt = self.optimizer
        if payload["optimizer"] != opt.name or payload["n_params"] != self.model.flat.numel():
            raise ValueError(f"optimizer state of {payload['optimizer']!r} over {payload['n_params']} parameters does not "
                             f"fit this engine ({opt.name!r}, {self.model.flat.numel()})")
        # (no flush here: the caller restored the weights already, and a replay of the OLD moments onto them would
        # perturb them -- resume_checkpoint flushes BEFORE it replaces the weights; moments and stamps are overwritten)
        dev = self.model.flat.device
        self._stats.view(torch.uint8).copy_(payload["stats"].to(dev))
        for attr in ("exp_avg", "exp_avg_sq"):
            buf = getattr(opt, attr)
            if buf is not None:
                buf.copy_(payload[attr].to(dev))
        lz = getattr(self, "_lazy", None)
        if lz is not None:
            if payload["lazy"] is not None:
                for k in ("stamp_u", "stamp_i", "scalars"):
                    lz[k].copy_(payload["lazy"][k].to(dev))
            else:       # written by the sweeping form: every row is current as of the restored clock
                self._lazy_mark_current()
            lz["dirty"] = False

    def bpr_loss(self, pos_scores, neg_scores):
        """torch_engine.py:92-106 on caller-supplied score tensors (utility, not the fused path)."""
        return -torch.mean(F.logsigmoid(pos_scores - neg_scores))

    def bce_loss(self, scores, ratings):
        """torch_engine.py:108-121 on caller-supplied score tensors (utility, not the fused path)."""
        return torch.nn.BCELoss()(scores, ratings)
