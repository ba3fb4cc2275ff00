"""Experiment: does the fold-stage Gram launch pack into the CUs that the fit-stage launch's
short (diagonal) workgroups leave early?  Runs fit() on one stream and the batched update on
another (separate objects = separate workspaces; timing only) and compares with back to back."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix, Partitioner

dev = torch.device("cuda:0")
N, K, M, P = 100000, 512, 16, 10
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.rand((N, K), dtype=torch.float64, device=dev, generator=g)
Y = torch.rand((N, M), dtype=torch.float64, device=dev, generator=g)
w = torch.rand((N,), dtype=torch.float64, device=dev, generator=g)
a = CVMatrix(copy=False, device=dev, lazy_fit=False); a.fit(X, Y, w)
b = CVMatrix(copy=False, device=dev, lazy_fit=False); b.fit(X, Y, w)
batch = b.prepare_folds(Partitioner(np.arange(N) % P))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def seq(reps):
    for _ in range(reps):
        a.fit(X, Y, w)
        o = b.training_XTX_XTY_batched(batch); del o

def par(reps):
    for _ in range(reps):
        with torch.cuda.stream(sa):
            a.fit(X, Y, w)
        with torch.cuda.stream(sb):
            o = b.training_XTX_XTY_batched(batch); del o

for name, fn in (("back to back", seq), ("two streams", par), ("back to back", seq), ("two streams", par)):
    fn(60); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(300); torch.cuda.synchronize()
    print(f"{name:14s} {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms per (fit + update)")

