import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cvmatrix_amd import CVMatrix
dev = torch.device("cuda")
def run(K, nv, nf, dt, flags):
    tdt = torch.float64 if dt is np.float64 else torch.float32
    g = torch.Generator(device=dev); g.manual_seed(1)
    N = 20000
    X = torch.rand((N, K), dtype=tdt, device=dev, generator=g); Y = torch.rand((N, 1), dtype=tdt, device=dev, generator=g)
    w = torch.rand((N,), dtype=tdt, device=dev, generator=g)
    m = CVMatrix(*flags, dtype=dt, copy=False, lazy_fit=False); m.fit(X, Y, w)
    b = m.prepare_folds([np.arange(i * nv, (i + 1) * nv) for i in range(nf)])
    o = m.training_XTX_XTY_batched(b); del o; torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); o = m.training_XTX_XTY_batched(b); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)); del o
    ms = float(np.median(ts)); s = np.dtype(dt).itemsize
    print(f"K={K} n={nv} folds={nf} {np.dtype(dt).name} flags={flags}: {ms:.3f} ms, written {nf*s*K*K/ms/1e9:.2f} TB/s")
for dt in (np.float64, np.float32):
    for flags in ((True,)*4, (True, True, False, False), (False,)*4):
        run(4096, 16, 48, dt, flags)
    run(4096, 1, 48, dt, (True,)*4)
    run(4096, 32, 48, dt, (True,)*4)
