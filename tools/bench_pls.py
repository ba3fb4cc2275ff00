"""Device PLS (cvm_pls_fit) timing on training matrices produced by the hot path.
Prints ms per call, folds/s, and the HBM view: XTX read once (slice resident in LDS) or once
per component (streamed)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix
from cvmatrix_amd.pls import pls_fit_batched, pls_plan


def run(name, N, K, M, P, A, dtype=np.float64, cpu=True, reps=7):
    tdt = torch.float64 if dtype is np.float64 else torch.float32
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.rand((N, K), dtype=tdt, device=dev, generator=g)
    Y = torch.rand((N, M), dtype=tdt, device=dev, generator=g)
    w = torch.rand((N,), dtype=tdt, device=dev, generator=g)
    m = CVMatrix(True, True, True, True, dtype=dtype, copy=False)
    m.fit(X, Y, w)
    nv = N // P
    folds = [np.arange(i * nv, (i + 1) * nv) for i in range(P)]
    batch = m.prepare_folds(folds)
    (XTX, XTY), stats = m.training_XTX_XTY_batched(batch)
    fit = pls_fit_batched(XTX, XTY, A)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fit = pls_fit_batched(XTX, XTY, A, check=False); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    plan = pls_plan(P, K, M, A, dtype)
    s = np.dtype(dtype).itemsize
    passes = 1 if plan["xtx_in_lds"] else A
    gb = P * (passes * K * K * s + A * K * M * s) / 1e9
    line = (f"{name:30s} F={P:6d} K={K:5d} M={M:3d} A={A:3d}: {ms:8.3f} ms  {P/ms*1e3:10.0f} folds/s  "
            f"{ms/A*1e3:7.1f} us/component  slices={plan['slices']:3d} lds_xtx={int(plan['xtx_in_lds'])}  {gb/ms*1e3:7.0f} GB/s")
    # the validation errors of all those models (cvm_pls_validation_sse): 2 n K A M flops per fold
    from cvmatrix_amd.pls import pls_validation_sse
    pls_validation_sse(m, batch, stats, fit.B); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0.record(); pls_validation_sse(m, batch, stats, fit.B); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms2 = float(np.median(ts))
    line += f"   validation sse {ms2:7.3f} ms ({2.0 * P * nv * K * A * M / ms2 / 1e9:5.1f} TFLOP/s)"
    if cpu:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle.ikpls_oracle import ikpls_fit
        a, b = XTX[0].cpu().numpy(), XTY[0].cpu().numpy()
        t0 = time.perf_counter(); ikpls_fit(a, b, A); t1 = time.perf_counter()
        line += f"   cpu oracle {1/(t1-t0):8.1f} folds/s"
    print(line, flush=True)


if __name__ == "__main__":
    run("C3 (headline shape)", 100000, 512, 16, 10, 20)
    run("C3, 30 components", 100000, 512, 16, 10, 30)
    run("C3 shape, 100 folds", 100000, 512, 16, 100, 20)
    run("C3 shape, 1000 folds", 100000, 512, 16, 1000, 20)
    run("K=500 M=10, 2000 folds", 100000, 500, 10, 2000, 20)
    run("C4 shape", 200000, 1024, 32, 64, 20)
    run("C5 shape fp32", 50000, 4096, 1, 20, 20, np.float32)
    run("C3 shape, M=48", 100000, 512, 48, 10, 20)
    run("C3 shape, M=64", 100000, 512, 64, 10, 20)
    run("K=128 M=64, 100 folds", 100000, 128, 64, 100, 20)
