"""Many small folds through mid_tile_kernel (large grids, several batches): sampled folds of the batched result
against one-fold calls (the direct small-fold kernels: another route).  python tools/big_folds_check.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix, _lib
import ctypes
lib = _lib.load()

dev = torch.device("cuda")
for (N, K, M, nv, dt) in ((1_000_000, 64, 2, 8, np.float64), (400_000, 256, 4, 16, np.float64), (600_000, 128, 4, 12, np.float32)):
    tdt = torch.float64 if dt is np.float64 else torch.float32
    g = torch.Generator(device=dev); g.manual_seed(N % 97)
    X = torch.rand((N, K), dtype=tdt, device=dev, generator=g)
    Y = torch.rand((N, M), dtype=tdt, device=dev, generator=g)
    w = torch.rand((N,), dtype=tdt, device=dev, generator=g)
    m = CVMatrix(copy=False, lazy_fit=False, dtype=dt); m.fit(X, Y, w)
    P = N // nv
    idx = np.arange(N).reshape(P, nv)
    folds = [idx[f] for f in range(P)]
    b = m.prepare_folds(folds)
    (bx, by), st = m.training_XTX_XTY_batched(b)
    torch.cuda.synchronize()
    # which kernels the batched call runs (the library's events by kind: 1 = Gram / tile kernel of the fold stage,
    # 2, 3 = the direct small-fold kernels)
    lib.cvm_timing_enable(1)
    o_ = m.training_XTX_XTY_batched(b); del o_
    torch.cuda.synchronize()
    ms4 = (ctypes.c_double * 4)(); n4 = (ctypes.c_int64 * 4)()
    lib.cvm_timing_read_kinds(ms4, n4)
    lib.cvm_timing_enable(0)
    kinds = {k: (int(n4[k]), round(ms4[k], 3)) for k in range(4) if n4[k]}
    worst = 0.0
    for f in (0, 1, P // 3, 16383, 16384, 16385, P - 2, P - 1):
        if f >= P:
            continue
        (ox, oy), so = m.training_XTX_XTY(folds[f])
        ex = float((bx[f] - ox).abs().max() / ox.abs().max())
        ey = float((by[f] - oy).abs().max() / oy.abs().max())
        es = max(float((a[f] - b).abs().max()) for a, b in zip(st, so) if a is not None)
        worst = max(worst, ex, ey, es)
    tol = 1e-12 if dt is np.float64 else 2e-5
    print(f"N={N} K={K} M={M} {P} folds of {nv} rows {np.dtype(dt).name}: launches by kind {kinds}; worst difference {worst:.2e}", "ok" if worst < tol else "FAIL")
    assert 1 in kinds and 3 not in kinds, "the batched call was expected to take the tile kernel"
    assert worst < tol
    del bx, by, st, X, Y, w, m
    torch.cuda.empty_cache()
