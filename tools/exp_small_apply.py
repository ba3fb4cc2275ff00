"""HBM regime ablation: the batched fold update at K=4096 for folds of 0 (no rows: only G - 0, finish, stores),
1, 4, 16 and 32 rows, float32 and float64; output TB/s.  CVM_LIB_PATH selects an experimental build.
    python tools/exp_small_apply.py [K]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda")
for dt, nf in ((np.float32, 48), (np.float64, 24)):
    tt = torch.float32 if dt is np.float32 else torch.float64
    g = torch.Generator(device=dev); g.manual_seed(1)
    N = 20000
    X = torch.rand((N, K), dtype=tt, device=dev, generator=g)
    Y = torch.rand((N, 1), dtype=tt, device=dev, generator=g)
    w = torch.rand((N,), dtype=tt, device=dev, generator=g)
    m = CVMatrix(dtype=dt, copy=False, lazy_fit=False); m.fit(X, Y, w)
    for nv in (0, 1, 4, 16, 32):
        folds = [np.arange(i * 32, i * 32 + nv) for i in range(nf)]
        b = m.prepare_folds(folds)
        o = m.training_XTX_XTY_batched(b); del o; torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(7):
            e0.record(); o = m.training_XTX_XTY_batched(b); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1)); del o
        ms = float(np.median(ts))
        out_bytes = nf * K * (K + 1) * np.dtype(dt).itemsize
        print(f"{np.dtype(dt).name} K={K} {nf} folds x {nv:2d} rows: {ms:7.3f} ms   outputs {out_bytes / ms / 1e9:6.2f} TB/s", flush=True)
    del X, Y, w, m
