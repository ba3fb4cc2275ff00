import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cvmatrix_amd import CVMatrix
dev = torch.device("cuda:0")
for (N, K, M) in ((6000000, 64, 2), (40000000, 8, 0)):
    g = torch.Generator(device=dev); g.manual_seed(3)
    X = torch.rand((N, K), dtype=torch.float64, device=dev, generator=g)
    Y = torch.rand((N, M), dtype=torch.float64, device=dev, generator=g) if M else None
    w = torch.rand((N,), dtype=torch.float64, device=dev, generator=g)
    m = CVMatrix(copy=False, device=dev, lazy_fit=False); m.fit(X, Y, w)
    Gr = (X * w[:, None]).T @ X
    print(N, K, "fit err", float((m.XTX - Gr).abs().max() / Gr.abs().max()))
    P = 4
    lab = torch.arange(N, device=dev) % P
    b = m.prepare_folds_from_labels(lab, P) if P <= 4096 else None
    if M:
        (xx, xy), st = m.training_XTX_XTY_batched(b)
    else:
        xx, st = m.training_XTX_batched(b)
    f = 1
    tr = lab != f
    Xt, wt = X[tr], w[tr]
    sw = wt.sum(); mu = (Xt * wt[:, None]).sum(0) / sw
    nz = (wt != 0).sum()
    var = ((Xt - mu) ** 2 * wt[:, None]).sum(0) / ((nz - 1) * sw / nz)
    Xs = (Xt - mu) / var.sqrt()
    ref = (Xs * wt[:, None]).T @ Xs
    print("   fold err", float((xx[f] - ref).abs().max() / ref.abs().max()), "mean err", float((st[0][f, 0] - mu).abs().max()))
    del X, Y, w, m, Gr, Xt, Xs, ref, xx
    torch.cuda.empty_cache()
