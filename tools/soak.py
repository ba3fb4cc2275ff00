"""Race screen outside the test-suite: the same step repeated many times must give the same
bits (deterministic reductions, counted LDS-DMA waits); several shapes, both element types."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix, Partitioner

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
# (the last three: mid-size folds with wide Y through the shared fused epilogue, leave-one-out through
#  the pipelined rows kernel, the lazy one-sweep fit)
for (N, K, M, P, dt, lazy) in ((100000, 512, 16, 10, torch.float64, False), (60000, 388, 34, 7, torch.float64, False),
                               (50000, 260, 0, 300, torch.float64, False), (80000, 516, 5, 6, torch.float32, False),
                               (30000, 130, 2, 3, torch.float64, False), (40000, 260, 40, 400, torch.float64, False),
                               (3000, 500, 10, 3000, torch.float64, False), (100000, 512, 16, 10, torch.float64, True),
                               # round 2: float32 fused epilogue, float32 rows kernel (leave-one-out), the one-call
                               # sweep with 16 folds and in float32, two folds
                               (60000, 516, 5, 600, torch.float32, False), (3000, 500, 10, 3000, torch.float32, False),
                               (64000, 388, 34, 16, torch.float64, True), (80000, 516, 3, 5, torch.float32, True),
                               (50000, 1028, 2, 2, torch.float64, True),
                               # round 4: mid_tile_kernel (folds of 100 / 33 / 12 rows), float64 and float32
                               (100000, 512, 16, 1000, torch.float64, False), (100000, 512, 16, 3000, torch.float32, False),
                               (24000, 1024, 4, 2000, torch.float64, False)):
    g = torch.Generator(device=dev); g.manual_seed(K)
    X = torch.rand((N, K), dtype=dt, device=dev, generator=g)
    Y = torch.rand((N, M), dtype=dt, device=dev, generator=g) if M else None
    w = torch.rand((N,), dtype=dt, device=dev, generator=g)
    m = CVMatrix(dtype=np.float64 if dt == torch.float64 else np.float32, copy=False, device=dev, lazy_fit=lazy)
    m.fit(X, Y, w)
    b = m.prepare_folds(Partitioner(np.arange(N) % P))
    ref = m.training_XTX_XTY_batched(b) if M else (m.training_XTX_batched(b),)
    ref = ref if M else ((ref[0][0], None), ref[0][1])
    bad = 0
    for i in range(reps):
        m.fit(X, Y, w)
        out = m.training_XTX_XTY_batched(b) if M else ((lambda r: ((r[0], None), r[1]))(m.training_XTX_batched(b)))
        (x, y), st = out
        (rx, ry), rst = ref
        ok = bool((x == rx).all()) and (y is None or bool((y == ry).all()))
        ok = ok and all(a is None or bool((a == c).all()) for a, c in zip(st, rst))
        bad += (not ok)
    print(f"N={N} K={K} M={M} P={P} {dt} lazy={lazy}: {reps} repetitions, {bad} differ")

# statistics-only calls (colstats_kernel: rows in flight, LDS-staged Y workgroups)
for (N, K, M, P, dt) in ((100000, 512, 16, 10, torch.float64), (50000, 4096, 1, 20, torch.float32), (90000, 260, 300, 1000, torch.float64)):
    g = torch.Generator(device=dev); g.manual_seed(K + 1)
    X = torch.rand((N, K), dtype=dt, device=dev, generator=g)
    Y = torch.rand((N, M), dtype=dt, device=dev, generator=g)
    w = torch.rand((N,), dtype=dt, device=dev, generator=g)
    m = CVMatrix(dtype=np.float64 if dt == torch.float64 else np.float32, copy=False, device=dev, lazy_fit=False)
    m.fit(X, Y, w)
    b = m.prepare_folds(Partitioner(np.arange(N) % P))
    ref = m.training_statistics_batched(b)
    bad = 0
    for i in range(reps):
        out = m.training_statistics_batched(b)
        bad += not all(bool((a == c).all()) for a, c in zip(out, ref))
    print(f"statistics N={N} K={K} M={M} P={P} {dt}: {reps} repetitions, {bad} differ")
