import contextlib, io, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
from test_ncf_gpu import make_engine, get_weights
from oracle import ncf_numpy as onc
U, I, L = 700, 500, 3
for E, B in ((32, 64), (32, 1000), (64, 100)):
    torch.manual_seed(E + L)
    eng = make_engine("NeuMFEngine", U, I, E, L, "adam", 1e-3, B)
    w = get_weights(eng)
    rng = np.random.default_rng(B)
    users, items = rng.integers(0, U, B), rng.integers(0, I, B)
    ratings = (rng.random(B) < 0.3).astype(np.float32)
    loss_ref, g_ref, _ = onc.ncf_grads(w, users, items, ratings, "neumf")
    loss, grads = eng.backward_only(torch.from_numpy(users), torch.from_numpy(items), torch.from_numpy(ratings))
    print("E", E, "B", B, "loss", loss, loss_ref)
    for k in g_ref:
        g = grads[k].cpu().numpy().reshape(g_ref[k].shape)
        sc = np.abs(g_ref[k]).max()
        print(f"   {k:32s} max err {np.abs(g - g_ref[k]).max() / sc:.3e}  |got|max/|ref|max {np.abs(g).max() / sc:.3f}  nonzero rows got {int((np.abs(g).reshape(g.shape[0], -1).max(1) > 0).sum())} ref {int((np.abs(g_ref[k]).reshape(g.shape[0], -1).max(1) > 0).sum())}")
    ws = eng.model.workspace(B)
    torch.cuda.synchronize()
    for l, (a, d) in enumerate(zip(ws["act"], ws["dact"])):
        print(f"   act[{l}] {tuple(a.shape)} |max| {float(a[:B].abs().max()):.3e} nz {float((a[:B] != 0).float().mean()):.3f}   dact[{l}] |max| {float(d[:B].abs().max()):.3e} nz {float((d[:B] != 0).float().mean()):.3f}")
