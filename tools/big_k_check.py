"""Wide matrices: K = 8192 and 16384 (G of 0.27 / 1 GB in float32, 2 GB in float64 at K = 16384), few
folds; the fold stage against a from-scratch torch computation of the centred / scaled training-set
matrices (float64 GEMM on the device)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix, Partitioner
dev = "cuda"
for (N, K, M, P, dt) in ((20000, 8192, 3, 5, torch.float32), (6000, 16384, 2, 2, torch.float32), (6000, 16384, 1, 3, torch.float64),
                         (3000, 8191, 2, 40, torch.float64)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.rand((N, K), dtype=dt, device=dev, generator=g)
    Y = torch.rand((N, M), dtype=dt, device=dev, generator=g)
    w = torch.rand((N,), dtype=dt, device=dev, generator=g)
    labels = np.arange(N) % P
    m = CVMatrix(dtype=np.float64 if dt == torch.float64 else np.float32)
    m.fit(X, Y, w)
    p = Partitioner(labels)
    (bx, by), st = m.training_XTX_XTY_batched(p)
    f = P - 1
    tr = torch.from_numpy(np.flatnonzero(labels != f)).to(dev)
    Xt, Yt, wt = X[tr].double(), Y[tr].double(), w[tr].double()
    sw = wt.sum()
    mux, muy = (wt[:, None] * Xt).sum(0) / sw, (wt[:, None] * Yt).sum(0) / sw
    Xc, Yc = Xt - mux, Yt - muy
    nz = (wt != 0).sum().double()
    div = (nz - 1) * sw / nz
    sdx = torch.sqrt((wt[:, None] * Xc * Xc).sum(0) / div); sdy = torch.sqrt((wt[:, None] * Yc * Yc).sum(0) / div)
    Xs, Ys = Xc / sdx, Yc / sdy
    rx = (Xs * wt[:, None]).T @ Xs
    ry = (Xs * wt[:, None]).T @ Ys
    ex = float((bx[f].double() - rx).abs().max() / rx.abs().max()); ey = float((by[f].double() - ry).abs().max() / ry.abs().max())
    es = float((st[1][f, 0].double() - sdx).abs().max() / sdx.abs().max())
    sym = bool((bx[f] == bx[f].T).all())
    print(f"N={N} K={K} M={M} P={P} {dt}: XTX err {ex:.2e} XTY err {ey:.2e} std err {es:.2e} symmetric {sym}", flush=True)
    bound = 1e-10 if dt == torch.float64 else 1e-3
    assert sym and ex <= bound and ey <= bound and es <= (1e-10 if dt == torch.float64 else 1e-4), (ex, ey, es, sym)
    del X, Y, w, m, bx, by, rx, ry, Xs, Ys, Xc, Yc, Xt
    torch.cuda.empty_cache()
