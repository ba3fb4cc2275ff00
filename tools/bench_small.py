"""Small-fold (HBM-bound) regime: folds of <= 32 rows go through small_stats/small_apply.
Reports folds/s and algorithmic GB/s (B_fold = s*n*(K+M+1) + 8n + 2*s*K*(K+M))."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix

def run(name, N, K, M, nv, nfolds, dtype, reps=5):
    tdt = torch.float64 if dtype is np.float64 else torch.float32
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.rand((N, K), dtype=tdt, device=dev, generator=g)
    Y = torch.rand((N, M), dtype=tdt, device=dev, generator=g)
    w = torch.rand((N,), dtype=tdt, device=dev, generator=g)
    m = CVMatrix(dtype=dtype, copy=False, lazy_fit=False)
    m.fit(X, Y, w)
    folds = [np.arange(i * nv, (i + 1) * nv) for i in range(nfolds)]
    b = m.prepare_folds(folds)
    out = m.training_XTX_XTY_batched(b); del out
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); out = m.training_XTX_XTY_batched(b); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)); del out
    ms = float(np.median(ts))
    s = np.dtype(dtype).itemsize
    bytes_alg = nfolds * (s * nv * (K + M + 1) + 8 * nv + 2 * s * K * (K + M))
    print(f"{name:34s} {nfolds:6d} folds x {nv:2d} rows: {ms:8.3f} ms  {nfolds/ms*1e3:10.0f} folds/s  "
          f"{bytes_alg/ms/1e6:7.0f} GB/s algorithmic ({bytes_alg/ms/1e6/8000:.2f} of 8 TB/s)")

if __name__ == "__main__":
    if os.environ.get("SMALL_F32"):
        run("K=500 M=10 fp32 LOOCV", 100000, 500, 10, 1, 4000, np.float32)
        run("K=500 M=10 fp32 n_v=2", 100000, 500, 10, 2, 4000, np.float32)
        run("K=250 M=10 fp32 LOOCV", 100000, 250, 10, 1, 8000, np.float32)
        run("K=1000 M=10 fp32 LOOCV", 100000, 1000, 10, 1, 1000, np.float32)
        run("K=900 M=10 fp32 LOOCV", 100000, 900, 10, 1, 1000, np.float32)
        sys.exit(0)
    run("K=4096 M=1 fp64 n_v=16", 20000, 4096, 1, 16, 24, np.float64)
    run("K=4096 M=1 fp32 n_v=16 (C5-hbm)", 20000, 4096, 1, 16, 48, np.float32)
    run("K=500 M=10 fp64 LOOCV", 100000, 500, 10, 1, 2000, np.float64)
    run("K=512 M=16 fp64 n_v=8", 100000, 512, 16, 8, 2000, np.float64)
    if os.environ.get("SMALL_EXTRA"):
        run("K=512 M=16 fp64 n_v=1", 100000, 512, 16, 1, 2000, np.float64)
        run("K=500 M=10 fp64 n_v=8", 100000, 500, 10, 8, 2000, np.float64)
        run("K=500 M=10 fp64 n_v=32", 100000, 500, 10, 32, 1000, np.float64)
        run("K=200 M=5 fp64 n_v=1", 100000, 200, 5, 1, 8000, np.float64)
        run("K=1000 M=4 fp64 n_v=4", 50000, 1000, 4, 4, 500, np.float64)
        run("K=500 M=10 fp32 n_v=1", 100000, 500, 10, 1, 4000, np.float32)

def write_ceiling():
    dev = torch.device("cuda")
    x = torch.empty(512 * 1024 * 1024, dtype=torch.float64, device=dev)   # 4 GiB
    y = torch.empty_like(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, fn, nbytes in (("fill (write only)", lambda: x.fill_(1.0), x.numel() * 8),
                             ("copy (read+write)", lambda: y.copy_(x), 2 * x.numel() * 8)):
        fn(); torch.cuda.synchronize()
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        print(f"{name:20s} {nbytes / e0.elapsed_time(e1) / 1e6:7.0f} GB/s")

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ceiling":
    write_ceiling()
