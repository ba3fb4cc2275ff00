import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix
dev = torch.device("cuda")
dt = np.float64 if (len(sys.argv) < 2 or sys.argv[1] == "f64") else np.float32
tdt = torch.float64 if dt is np.float64 else torch.float32
K, nv, nf, N = 4096, 16, 48, 20000
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.rand((N, K), dtype=tdt, device=dev, generator=g); Y = torch.rand((N, 1), dtype=tdt, device=dev, generator=g)
w = torch.rand((N,), dtype=tdt, device=dev, generator=g)
m = CVMatrix(dtype=dt, copy=False, lazy_fit=False); m.fit(X, Y, w)
b = m.prepare_folds([np.arange(i * nv, (i + 1) * nv) for i in range(nf)])
for _ in range(3):
    o = m.training_XTX_XTY_batched(b); del o
torch.cuda.synchronize()
