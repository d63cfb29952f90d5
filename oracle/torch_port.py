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


def _make_optimizer(params, optimizer, lr):
    if optimizer == "sgd":
        return torch.optim.SGD(params, lr=lr)
    if optimizer == "adam":
        return torch.optim.Adam(params, lr=lr)
    if optimizer == "rmsprop":
        return torch.optim.RMSprop(params, lr=lr)
    raise ValueError(optimizer)


class _Port:
    """Leaf tensors keyed like the reference's state_dict + a stock torch optimizer over them."""

    def __init__(self, weights, optimizer, lr):
        self.w = {k: torch.as_tensor(v, dtype=torch.float32).clone().requires_grad_(True)
                  for k, v in weights.items()}
        self.opt = _make_optimizer(list(self.w.values()), optimizer, lr)

    def numpy_weights(self):
        return {k: v.detach().numpy().copy() for k, v in self.w.items()}


class TorchPGMFPort(_Port):
    """PairwiseGMFEngine.train_single_batch as ATen ops (models/pairwise_gmf.py:48-62, 82-116, 144-158):
    relu(Linear(u * i)), -log(sigmoid(.) + 1e-12), + lambda * ||v||, clip_grad_norm_, optimizer step.
    Pinned by tests/golden/pgmf_*.npz."""

    def __init__(self, weights, optimizer="adam", lr=1e-4, l2_lambda=1e-4, grad_clip=5.0):
        super().__init__(weights, optimizer, lr)
        self.l2_lambda, self.grad_clip = l2_lambda, grad_clip

    def step(self, batch):
        users, pos, neg = (torch.as_tensor(x, dtype=torch.int64) for x in batch)
        w = self.w
        self.opt.zero_grad()
        u = F.embedding(users, w["user_memory.weight"])
        score = lambda items: F.relu(F.linear(u * F.embedding(items, w["item_memory.weight"]), w["v.weight"]))  # noqa: E731
        loss = torch.mean(-1 * torch.log(torch.sigmoid(score(pos) - score(neg)) + 1e-12))
        loss = loss + self.l2_lambda * torch.sqrt(w["v.weight"].pow(2).sum())
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(w.values()), self.grad_clip)
        self.opt.step()
        return loss.item()


class TorchT2VPort(_Port):
    """Triple2vecEngine.train_single_batch as ATen ops (models/triple2vec.py:36-92, 115-124) with
    item_emb2 aliased to item_emb1 (use_bias = n_neg) and both negative item rows gathered by neg_i_2.
    Weights: user_emb / item_emb1 / user_bias / item_bias.  Pinned by tests/golden/t2v_*.npz."""

    def __init__(self, weights, batch_size, optimizer="adam", lr=5e-4):
        super().__init__({k: v for k, v in weights.items() if k != "item_emb2.weight"}, optimizer, lr)
        self.batch_size = batch_size

    def step(self, batch):
        pu, p1, p2, nu, n1, n2 = (torch.as_tensor(x, dtype=torch.int64) for x in batch)
        w = self.w
        E, bias_u, bias_i = w["item_emb1.weight"], w["user_bias.weight"], w["item_bias.weight"]
        self.opt.zero_grad()
        eu, e1, e2 = F.embedding(pu, w["user_emb.weight"]), F.embedding(p1, E), F.embedding(p2, E)
        ru, r1, r2 = F.embedding(nu, w["user_emb.weight"]), F.embedding(n2, E), F.embedding(n2, E)

        def part(center, context, bias_pos, neg_rows, bias_neg):
            pos_s = F.logsigmoid(torch.sum(center * context, dim=1) + bias_pos.squeeze())
            neg_s = F.logsigmoid(-1 * (torch.bmm(neg_rows, center.unsqueeze(2)).squeeze() + bias_neg.squeeze()))
            return -1 * (torch.sum(pos_s) + torch.sum(neg_s))

        total = (part(eu, e1 + e2, F.embedding(pu, bias_u), ru, F.embedding(nu, bias_u))
                 + part(e1, eu + e2, F.embedding(p1, bias_i), r1, F.embedding(n1, bias_i))
                 + part(e2, eu + e1, F.embedding(p2, bias_i), r2, F.embedding(n2, bias_i)))
        loss = total / (3 * self.batch_size)
        loss.backward()
        self.opt.step()
        return loss.item()


class TorchNGCFPort(_Port):
    """NGCFEngine.train_single_batch as ATen ops (models/ngcf.py:48-80, 118-149, 172-199) on a torch sparse
    norm_adj; nn.Dropout draws from the global CPU generator as in the reference.  Pinned by
    tests/golden/ngcf_*.npz."""

    def __init__(self, weights, norm_adj, mess_dropout, decay, batch_size, optimizer="adam", lr=0.05):
        super().__init__(weights, optimizer, lr)
        self.adj, self.drop, self.decay, self.batch_size = norm_adj, list(mess_dropout), decay, batch_size
        self.n_layers = sum(1 for k in weights if k.startswith("GC_weights.") and k.endswith(".weight"))
        self.training = True

    def forward(self):
        w = self.w
        ego = torch.cat((w["user_embedding.weight"], w["item_embedding.weight"]), dim=0)
        outs = [ego]
        for l in range(self.n_layers):
            side = torch.sparse.mm(self.adj, ego)
            s = F.leaky_relu(F.linear(side, w[f"GC_weights.{l}.weight"], w[f"GC_weights.{l}.bias"]))
            b = F.leaky_relu(F.linear(ego * side, w[f"Bi_weights.{l}.weight"], w[f"Bi_weights.{l}.bias"]))
            ego = F.dropout(s + b, self.drop[l], self.training)
            outs.append(F.normalize(ego, p=2, dim=1))
        return torch.cat(outs, dim=1)

    def step(self, batch):
        users, pos, neg = (torch.as_tensor(x, dtype=torch.int64) for x in batch)
        n_users = self.w["user_embedding.weight"].shape[0]
        self.opt.zero_grad()
        allv = self.forward()
        u, p, n = allv[users], allv[n_users + pos], allv[n_users + neg]
        reg = (0.5 * (u ** 2).sum() + 0.5 * (p ** 2).sum() + 0.5 * (n ** 2).sum()) / self.batch_size
        loss = -torch.mean(F.logsigmoid((u * p).sum(1) - (u * n).sum(1))) + self.decay * reg
        loss.backward()
        self.opt.step()
        return loss.item()


class TorchNeuMFPort(_Port):
    """NeuMFEngine.train_single_batch as ATen ops (models/ncf.py:52-71, 100-120): four embedding gathers, the tower
    walked module by module with a ReLU after EVERY module of ``fc_layers`` (Dropout, Linear and ReLU alike -- so the
    concatenated embeddings are rectified before the first Linear, quirk Q7), GMF product, ``affine_output``,
    sigmoid, BCELoss(mean), backward, optimizer step.  Weights keyed like the reference's state_dict
    (``fc_layers.{1,4,7,..}``).  Pinned by tests/golden/ncf_neumf_*.npz (tests/test_torch_ports.py)."""

    def __init__(self, weights, optimizer="adam", lr=1e-3, dropout=0.0):
        super().__init__(weights, optimizer, lr)
        self.layers = sorted(int(k.split(".")[1]) for k in weights if k.startswith("fc_layers.") and k.endswith(".weight"))
        self.dropout, self.training = dropout, True

    def forward(self, users, items):
        w = self.w
        mlp = torch.cat([F.embedding(users, w["embedding_user_mlp.weight"]),
                         F.embedding(items, w["embedding_item_mlp.weight"])], dim=-1)
        mf = torch.mul(F.embedding(users, w["embedding_user_mf.weight"]), F.embedding(items, w["embedding_item_mf.weight"]))
        for idx in self.layers:     # module idx - 1 is the Dropout in front of Linear idx, idx + 1 the ReLU behind it
            mlp = torch.relu(F.dropout(mlp, self.dropout, self.training))
            mlp = torch.relu(F.linear(mlp, w[f"fc_layers.{idx}.weight"], w[f"fc_layers.{idx}.bias"]))
            mlp = torch.relu(torch.relu(mlp))
        logits = F.linear(torch.cat([mlp, mf], dim=-1), w["affine_output.weight"], w["affine_output.bias"])
        return torch.sigmoid(logits)

    def step(self, batch):
        users, items = (torch.as_tensor(x, dtype=torch.int64) for x in batch[:2])
        ratings = torch.as_tensor(batch[2], dtype=torch.float32)
        self.opt.zero_grad()
        loss = torch.nn.BCELoss()(self.forward(users, items).view(-1), ratings)
        loss.backward()
        self.opt.step()
        return loss.item()


class TorchLightGCNPort(_Port):
    """LightGCNEngine.train_single_batch as ATen ops (models/lightgcn.py:27-38, 46-78, 119-152, 171-191) on a torch
    sparse norm_adj: edge dropout drawn with ``torch.rand(nnz)`` from the global CPU generator like the reference
    (keep where int(rand + keep_prob) != 0, scale 1 / keep_prob), L x ``torch.sparse.mm``, mean over the layer
    stack, softplus BPR + decay * L2 of the layer-0 rows, backward through the sparse products, optimizer step.
    Pinned by tests/golden/lightgcn_*.npz (tests/test_torch_ports.py)."""

    def __init__(self, weights, norm_adj, n_layers, keep_prob, decay, optimizer="adam", lr=0.05):
        super().__init__(weights, optimizer, lr)
        self.adj, self.n_layers, self.keep, self.decay = norm_adj.coalesce(), n_layers, keep_prob, decay
        self.training = True

    def propagate(self):
        w, adj = self.w, self.adj
        if self.training:
            keep = (torch.rand(adj.values().numel()) + self.keep).int().bool()
            adj = torch.sparse_coo_tensor(adj.indices()[:, keep], adj.values()[keep] / self.keep, adj.size())
        e = torch.cat((w["user_embedding.weight"], w["item_embedding.weight"]), dim=0)
        embs = [e]
        for _ in range(self.n_layers):
            e = torch.sparse.mm(adj, e)
            embs.append(e)
        out = torch.mean(torch.stack(embs, dim=1), dim=1)
        n_users = w["user_embedding.weight"].shape[0]
        return out[:n_users], out[n_users:]

    def step(self, batch):
        users, pos, neg = (torch.as_tensor(x, dtype=torch.int64) for x in batch)
        w = self.w
        self.opt.zero_grad()
        ua, ia = self.propagate()
        u, p, n = ua[users], ia[pos], ia[neg]
        u0 = F.embedding(users, w["user_embedding.weight"])
        p0, n0 = F.embedding(pos, w["item_embedding.weight"]), F.embedding(neg, w["item_embedding.weight"])
        reg = 0.5 * (u0.norm(2).pow(2) + p0.norm(2).pow(2) + n0.norm(2).pow(2)) / float(len(users)) * self.decay
        loss = torch.mean(F.softplus(torch.sum(u * n, dim=1) - torch.sum(u * p, dim=1))) + reg
        loss.backward()
        self.opt.step()
        return loss.item()
