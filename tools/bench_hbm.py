"""HBM regime, measured like the headline: >= 30 ms of pre-warm, then R calls back to back between one pair
of events.  Outputs TB/s of bytes that must move (outputs + rows once per fold, G and H once per launch).
    python tools/bench_hbm.py [quick]
CVM_SMALL_TILE=0 -> the round-3 kernels; 1 (default) -> small_tile_kernel; 2 -> also instead of the rows kernel."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix

dev = torch.device("cuda")


def run(name, N, K, M, nv, nfolds, dtype, flags=(True,) * 4, weighted=True, R=8):
    tdt = torch.float64 if dtype is np.float64 else torch.float32
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.rand((N, K), dtype=tdt, device=dev, generator=g)
    Y = torch.rand((N, M), dtype=tdt, device=dev, generator=g)
    w = torch.rand((N,), dtype=tdt, device=dev, generator=g) if weighted else None
    m = CVMatrix(*flags, dtype=dtype, copy=False, lazy_fit=False)
    m.fit(X, Y, w)
    b = m.prepare_folds([np.arange(i * nv, (i + 1) * nv) for i in range(nfolds)])
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        o = m.training_XTX_XTY_batched(b); del o
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _k in range(R):
            o = m.training_XTX_XTY_batched(b); del o
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / R)
    ms = float(np.median(ts))
    s = np.dtype(dtype).itemsize
    bts = nfolds * (s * nv * (K + M + 1) + 8 * nv + s * K * (K + M)) + s * K * (K + M)
    print(f"{name:36s} {nfolds:5d} folds x {nv:2d} rows: {ms:7.3f} ms {nfolds / ms * 1e3:10.0f} folds/s "
          f"{bts / ms / 1e9:6.2f} TB/s = {bts / ms / 1e9 / 8:.3f} of 8 TB/s", flush=True)
    del X, Y, w, m, b


def fill():
    x = torch.empty(3 * 1024 ** 3 // 4, dtype=torch.float32, device=dev)
    for _ in range(10):
        x.fill_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        x.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    print(f"torch fill of 3 GiB: {x.numel() * 4 * 8 / e0.elapsed_time(e1) / 1e9:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    print("CVM_SMALL_TILE =", os.environ.get("CVM_SMALL_TILE", "(default)"), " CVM_SMALL_FPB =", os.environ.get("CVM_SMALL_FPB", "-"))
    fill()
    if len(sys.argv) > 1 and sys.argv[1] == "resident":     # the shapes of the round-6 resident route (CVM_RESIDENT=1: wherever the shape allows; 0: never; default: K >= 4096 and >= 32 folds)
        print("CVM_RESIDENT =", os.environ.get("CVM_RESIDENT", "(default)"))
        run("C5-hbm K=4096 M=1 f32 n=16", 20000, 4096, 1, 16, 48, np.float32)
        run("K=4096 M=1 f32 n=8", 20000, 4096, 1, 8, 48, np.float32)
        run("K=4096 M=1 f32 n=1", 20000, 4096, 1, 1, 48, np.float32)
        run("K=4096 M=1 f32 n=32", 20000, 4096, 1, 32, 48, np.float32)
        run("K=4096 M=1 f32 n=24", 20000, 4096, 1, 24, 48, np.float32)
        if len(sys.argv) > 2 and sys.argv[2] == "first":
            sys.exit(0)
        run("K=4096 M=1 f32 n=16 12 folds", 20000, 4096, 1, 16, 12, np.float32)
        run("K=4096 M=1 f32 n=16 24 folds", 20000, 4096, 1, 16, 24, np.float32)
        run("K=4096 M=1 f32 n=16 32 folds", 20000, 4096, 1, 16, 32, np.float32)
        run("K=4096 M=1 f32 n=16 160 folds", 20000, 4096, 1, 16, 160, np.float32)
        if len(sys.argv) > 2 and sys.argv[2] == "wide":
            run("K=8192 M=1 f32 n=16 40 folds", 20000, 8192, 1, 16, 40, np.float32, R=4)
            run("K=8192 M=1 f32 n=8 40 folds", 20000, 8192, 1, 8, 40, np.float32, R=4)
            run("K=2048 M=1 f32 n=16 80 folds", 20000, 2048, 1, 16, 80, np.float32)
            run("K=2048 M=1 f32 n=16 160 folds", 20000, 2048, 1, 16, 160, np.float32)
            run("K=2048 M=1 f32 n=8 160 folds", 20000, 2048, 1, 8, 160, np.float32)
            run("K=2048 M=1 f32 n=16 240 folds", 20000, 2048, 1, 16, 240, np.float32)
            sys.exit(0)
        run("K=3072 M=1 f32 n=16", 20000, 3072, 1, 16, 80, np.float32)
        run("K=2048 M=8 f32 n=16 no centre/scale", 20000, 2048, 8, 16, 400, np.float32, flags=(False,) * 4)
        run("K=2048 M=8 f32 n=16", 20000, 2048, 8, 16, 400, np.float32)
        run("K=2048 M=1 f32 n=4", 20000, 2048, 1, 4, 400, np.float32)
        run("K=1024 M=1 f32 n=16", 50000, 1024, 1, 16, 1000, np.float32)
        run("K=1024 M=1 f32 n=4", 50000, 1024, 1, 4, 1000, np.float32)
        sys.exit(0)
    run("C5-hbm K=4096 M=1 f32 n=16", 20000, 4096, 1, 16, 48, np.float32)
    run("K=4096 M=1 f64 n=16", 20000, 4096, 1, 16, 48, np.float64)
    run("LOOCV K=500 M=10 f64 n=1", 100000, 500, 10, 1, 2000, np.float64)
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        sys.exit(0)
    run("K=4096 M=1 f32 n=32", 20000, 4096, 1, 32, 48, np.float32)
    run("K=4096 M=1 f64 n=32", 20000, 4096, 1, 32, 48, np.float64)
    run("K=4096 M=1 f32 n=1", 20000, 4096, 1, 1, 48, np.float32)
    run("K=4096 M=1 f64 n=4", 20000, 4096, 1, 4, 48, np.float64)
    run("K=512 M=16 f64 n=8", 100000, 512, 16, 8, 2000, np.float64)
    run("K=512 M=16 f64 n=1", 100000, 512, 16, 1, 2000, np.float64)
    run("K=500 M=10 f32 n=1", 100000, 500, 10, 1, 4000, np.float32)
    run("K=1000 M=4 f64 n=4", 50000, 1000, 4, 4, 500, np.float64)
    run("K=2048 M=8 f64 n=16 unweighted", 20000, 2048, 8, 16, 200, np.float64, weighted=False)
    run("K=2048 M=8 f32 n=16 no centre/scale", 20000, 2048, 8, 16, 400, np.float32, flags=(False,) * 4)
