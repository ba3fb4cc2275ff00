"""Where the direct small-fold kernels stop paying: folds of n rows (n > 32) through the direct
route (CVM_SMALL_MAXN=128, set by the caller) or the fused Gram route (default), same box.
    python tools/exp_small_limit.py [K M dtype]       prints ms and output TB/s per fold size"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix

K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dt = np.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else np.float64
tdt = torch.float64 if dt is np.float64 else torch.float32
N = 100000
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.rand((N, K), dtype=tdt, device=dev, generator=g)
Y = torch.rand((N, M), dtype=tdt, device=dev, generator=g)
w = torch.rand((N,), dtype=tdt, device=dev, generator=g)
m = CVMatrix(copy=False, lazy_fit=False, dtype=dt); m.fit(X, Y, w)
es = np.dtype(dt).itemsize
print(f"CVM_SMALL_MAXN={os.environ.get('CVM_SMALL_MAXN', '(default)')} K={K} M={M} {np.dtype(dt).name}")
NVS = [int(v) for v in os.environ['NVS'].split(',')] if os.environ.get('NVS') else (16, 32, 33, 40, 48, 64, 80, 100, 128, 160)
for nv in NVS:
    nf = min(N // nv, max(1, int(6e9 // (K * (K + M) * es))))
    folds = [np.arange(f * nv, (f + 1) * nv) for f in range(nf)]
    b = m.prepare_folds(folds)
    o = m.training_XTX_XTY_batched(b); del o; torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record(); o = m.training_XTX_XTY_batched(b); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)); del o
    ms = float(np.median(ts))
    print(f"  n_val={nv:4d} folds {nf:5d}: {ms:8.3f} ms  {nf/ms*1e3:10.0f} folds/s  outputs {nf*K*(K+M)*es/ms/1e9:6.2f} TB/s", flush=True)
