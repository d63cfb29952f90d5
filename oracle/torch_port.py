"""ORACLE (test infrastructure — never imported by the product path).

PyTorch-CPU port of the reference's MF training step: the SAME ATen op sequence the reference
issues (embedding gathers, mul/sum, sigmoid, logsigmoid / BCELoss, autograd backward producing
dense gradients, torch.optim.{SGD,Adam,RMSprop}.step with only ``lr`` set), written functionally
over explicit weight tensors.  It serves two purposes:

* a second, independent check of the HIP path (it shares no code with oracle/mf_numpy.py);
* the ``cpu_baseline`` leg of bench.py ("kind": "port"): what the reference's own CPU path costs
  on the GPU box's host cores (multi-threaded ATen, all cores).

Reference lines restated (relative to /root/reference/beta_rec/): models/mf.py:32-55 (forward),
models/mf.py:92-119 (train_single_batch), models/torch_engine.py:23-39 (optimizers),
models/torch_engine.py:92-121 (losses).  Pinned by tests/golden/mf_*.npz (captured from the real
reference by oracle/gen_golden.py).
"""
import torch
import torch.nn.functional as F

KEYS = ("global_bias", "user_emb.weight", "item_emb.weight", "user_bias.weight", "item_bias.weight")


class TorchMFPort:
    """Weights as leaf tensors + a stock torch optimizer over them."""

    def __init__(self, weights, optimizer="sgd", lr=0.05, loss="bpr", reg=0.0):
        self.w = {
            k: torch.as_tensor(weights[k], dtype=torch.float32).clone().requires_grad_(True)
            for k in KEYS
        }
        params = [self.w[k] for k in KEYS]
        if optimizer == "sgd":
            self.opt = torch.optim.SGD(params, lr=lr)
        elif optimizer == "adam":
            self.opt = torch.optim.Adam(params, lr=lr)
        elif optimizer == "rmsprop":
            self.opt = torch.optim.RMSprop(params, lr=lr)
        else:
            raise ValueError(optimizer)
        self.loss = loss
        self.reg = reg

    def forward(self, users, items):
        """models/mf.py:32-55."""
        w = self.w
        u = F.embedding(users, w["user_emb.weight"])
        bu = F.embedding(users, w["user_bias.weight"])
        i = F.embedding(items, w["item_emb.weight"])
        bi = F.embedding(items, w["item_bias.weight"])
        scores = torch.sigmoid(
            torch.sum(torch.mul(u, i).squeeze(), dim=1) + bu.squeeze() + bi.squeeze()
            + w["global_bias"]
        )
        reg = ((u ** 2).sum() + (i ** 2).sum() + (bu ** 2).sum() + (bi ** 2).sum()) / u.size()[0]
        return scores, reg

    def grads(self, batch):
        """zero_grad + forward + backward; returns (loss, reg) tensors, grads left in .grad."""
        self.opt.zero_grad()
        if self.loss == "bpr":
            users, pos, neg = batch
            ps, pr = self.forward(users, pos)
            ns, nr = self.forward(users, neg)
            loss = -torch.mean(F.logsigmoid(ps - ns))
            reg = pr + nr
        else:
            users, items, ratings = batch
            s, reg = self.forward(users, items)
            loss = torch.nn.BCELoss()(s, ratings)
        (loss + self.reg * reg).backward()
        return loss, reg

    def step(self, batch):
        """models/mf.py:92-119."""
        loss, reg = self.grads(batch)
        self.opt.step()
        return loss.item(), reg.item()

    def numpy_weights(self):
        return {k: v.detach().numpy().copy() for k, v in self.w.items()}

    def numpy_grads(self):
        return {
            k: (v.grad.numpy().copy() if v.grad is not None else torch.zeros_like(v).numpy())
            for k, v in self.w.items()
        }

    def predict(self, users, items):
        with torch.no_grad():
            return self.forward(users, items)[0]
