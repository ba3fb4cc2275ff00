"""Diagnostic: statistics-only fold stage (colstats_kernel) at a few shapes; run under
rocprofv3 --kernel-trace --stats for kernel-only durations."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvmatrix_amd._lib as L
SHAPES = ((100000, 512, 16, 10, torch.float64), (200000, 4096, 1, 20, torch.float32),
          (100000, 512, 16, 1000, torch.float64))
if len(sys.argv) > 1:            # indices of the shapes to run, e.g. "0,1"
    SHAPES = tuple(SHAPES[int(i)] for i in sys.argv[1].split(","))
from cvmatrix_amd import CVMatrix, Partitioner

dev = torch.device("cuda:0")
for (N, K, M, P, dt) in SHAPES:
    g = torch.Generator(device=dev); g.manual_seed(0)
    X = torch.rand((N, K), dtype=dt, device=dev, generator=g)
    Y = torch.rand((N, M), dtype=dt, device=dev, generator=g)
    w = torch.rand((N,), dtype=dt, device=dev, generator=g)
    m = CVMatrix(dtype=np.float64 if dt == torch.float64 else np.float32, copy=False, device=dev, lazy_fit=False)
    m.fit(X, Y, w)
    b = m.prepare_folds(Partitioner(np.arange(N) % P))
    m.training_statistics_batched(b); torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        m.training_statistics_batched(b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    es = X.element_size()
    bts = N * (es * (K + M + 1) + 8)
    print(f"N={N} K={K} M={M} P={P} {dt}: {ms:.4f} ms/call (host+device, back to back) "
          f"{bts / ms / 1e6:.0f} GB/s of {bts / 1e6:.0f} MB", flush=True)
    del X, Y, w, m, b
