"""Fold-size sweep at the C3 data shape (the reference's benchmark varies P the same way,
benchmarks/benchmark.py:239): folds/s of the batched update for P = N / n_val."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix

N, K, M = 100000, 512, 16
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.rand((N, K), dtype=torch.float64, device=dev, generator=g)
Y = torch.rand((N, M), dtype=torch.float64, device=dev, generator=g)
w = torch.rand((N,), dtype=torch.float64, device=dev, generator=g)
m = CVMatrix(copy=False, lazy_fit=False); m.fit(X, Y, w)
PS = [int(p) for p in os.environ["FOLD_PS"].split(",")] if os.environ.get("FOLD_PS") else (3, 5, 10, 30, 100, 300, 1000, 2000, 3000)
for P in PS:
    nv = N // P
    nf = min(P, max(1, int(8e9 // (K * (K + M) * 8))))      # cap the output at 8 GB
    if os.environ.get("FOLD_CONTIG"):      # diagnostic: contiguous validation rows instead of the reference's f, f + P, ...
        folds = [np.arange(f * nv, (f + 1) * nv) for f in range(nf)]
    else:
        folds = [np.arange(f, N, P)[:nv] for f in range(nf)]
    b = m.prepare_folds(folds)
    o = m.training_XTX_XTY_batched(b); del o; torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record(); o = m.training_XTX_XTY_batched(b); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)); del o
    ms = float(np.median(ts))
    fl = nf * nv * (K * (K + 1) + 2.0 * K * M)
    print(f"P={P:6d} n_val={nv:6d} folds timed {nf:5d}: {ms:9.3f} ms  {nf/ms*1e3:10.0f} folds/s  {fl/ms/1e9:6.1f} TFLOP/s(sym)")
