"""End-to-end BPR-MF on the device: what beta_rec.recommenders.MatrixFactorization.train() does per
run, with every stage on the MI355X.

    python examples/mf_end_to_end.py [--epochs 5] [--interactions 1000000]

Stages (reference file:line -> here):
  * training set      data/base_data.py:218-253 instance_bpr_loader   -> beta_recsys_amd.data.instance_bpr_loader
                      (device-side negative sampling + device batcher)
  * epoch             models/mf.py:121-139 train_an_epoch              -> MFEngine.train_an_epoch (one fused kernel per step)
  * validation        core/eval_engine.py:231-274 predict + :49-87 evaluate -> beta_recsys_amd.eval.predict_device / evaluate
                      (leave-one-out: 1 held-out positive + 100 sampled negatives per user)
The data is synthetic and MovieLens-1M-shaped (6040 users x 3706 items, Zipf item popularity); no
dataset is downloaded.  Prints one JSON line per epoch and a summary line.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def synth_interactions(n_users, n_items, n, seed):
    """Unique (user, item) pairs: uniform users, Zipf(1.0) items over a random permutation."""
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, n_items + 1)
    users = rng.integers(0, n_users, int(n * 1.3))
    items = rng.permutation(n_items)[rng.choice(n_items, users.size, p=p / p.sum())]
    key = np.unique(users.astype(np.int64) * n_items + items)
    key = key[rng.permutation(key.size)][:n]
    return key // n_items, key % n_items


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=6040)
    ap.add_argument("--items", type=int, default=3706)
    ap.add_argument("--interactions", type=int, default=1_000_000)
    ap.add_argument("--emb-dim", type=int, default=64)
    ap.add_argument("--batch-size", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--optimizer", default="adam")
    ap.add_argument("--lr", type=float, default=0.005)
    ap.add_argument("--eval-negatives", type=int, default=100)
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X (torch.cuda.is_available() is False)")
    import pandas as pd

    import beta_recsys_amd as hp

    torch.manual_seed(2020)
    U, I = args.users, args.items
    users, items = synth_interactions(U, I, args.interactions, seed=1)
    # leave-one-out: the last-listed interaction of every user is held out for validation
    order = np.lexsort((np.arange(users.size), users))
    last = np.r_[users[order][1:] != users[order][:-1], True]
    held = order[last]
    train_mask = np.ones(users.size, dtype=bool)
    train_mask[held] = False
    train = pd.DataFrame({"col_user": users[train_mask], "col_item": items[train_mask],
                          "col_rating": np.ones(int(train_mask.sum()), dtype=np.float32)})
    data = types.SimpleNamespace(train=train, n_users=U, n_items=I)

    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        loader = hp.data.instance_bpr_loader(data, args.batch_size, "cuda:0")
        # validation frame: held-out positive + sampled negatives per user (sampled against ALL of the
        # user's interactions, as the reference's leave-one-out splitter does)
        neg = hp.data.sample_negatives(users, items, U, I, k=args.eval_negatives, device="cuda:0")[torch.as_tensor(held).cuda()]
        torch.cuda.synchronize()
    t_data = time.perf_counter() - t0
    hu = torch.as_tensor(users[held]).cuda()
    val_users = hu.repeat_interleave(args.eval_negatives + 1)
    val_items = torch.cat([torch.as_tensor(items[held]).cuda()[:, None], neg], 1).reshape(-1)
    val_ratings = torch.tensor([1.0] + [0.0] * args.eval_negatives, device="cuda:0").repeat(hu.numel())
    val = {"col_user": val_users, "col_item": val_items.cpu().numpy(), "col_rating": val_ratings}
    val_np = {"col_user": val_users.cpu().numpy(), "col_item": val["col_item"]}

    cfg = {"model": dict(n_users=U, n_items=I, emb_dim=args.emb_dim, device_str="cuda:0", optimizer=args.optimizer,
                         lr=args.lr, batch_size=args.batch_size, loss="bpr"),
           "system": {"run_dir": "/tmp/hiprec_example_runs"}}
    with contextlib.redirect_stdout(io.StringIO()):
        eng = hp.MFEngine(cfg)
    print(json.dumps({"stage": "data", "train_rows": int(train_mask.sum()), "val_rows": int(val_users.numel()),
                      "seconds": round(t_data, 3)}), flush=True)
    history = []
    for epoch in range(args.epochs):
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            eng.train_an_epoch(loader, epoch)
        torch.cuda.synchronize()
        t_train = time.perf_counter() - t0
        t0 = time.perf_counter()
        scores = hp.eval.predict_device(val_np, eng.model)
        metrics = hp.eval.evaluate(val, scores, ["ndcg", "recall"], [5, 10, 20])
        t_eval = time.perf_counter() - t0
        st = eng.epoch_stats()
        row = {"epoch": epoch, "loss": round(st.loss_sum / max(len(loader), 1), 5), "train_s": round(t_train, 4),
               "eval_s": round(t_eval, 4), "triples_per_s": round(len(train) / t_train),
               **{k: round(v, 4) for k, v in metrics.items()}}
        history.append(row)
        print(json.dumps(row), flush=True)
    print(json.dumps({"stage": "summary", "epochs": args.epochs,
                      "mean_epoch_s": round(float(np.mean([h["train_s"] + h["eval_s"] for h in history])), 4),
                      "ndcg@10_first": history[0]["ndcg@10"], "ndcg@10_last": history[-1]["ndcg@10"]}), flush=True)
    return history


if __name__ == "__main__":
    main()
