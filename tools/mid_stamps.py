"""Where a wave of mid_tile_kernel spends its cycles (build with -DCVM_STAMPS):
   python tools/mid_stamps.py tools/libcvmhip_stamps.so [P]"""
import ctypes, os, sys
import numpy as np, torch
os.environ["CVM_LIB_PATH"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix, _lib
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
N, K, M = 100000, 512, 16
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.rand((N, K), dtype=torch.float64, device=dev, generator=g)
Y = torch.rand((N, M), dtype=torch.float64, device=dev, generator=g)
w = torch.rand((N,), dtype=torch.float64, device=dev, generator=g)
m = CVMatrix(copy=False, lazy_fit=False); m.fit(X, Y, w)
nv = N // P
b = m.prepare_folds([np.arange(f, N, P)[:nv] for f in range(P)])
for _ in range(3):
    o = m.training_XTX_XTY_batched(b); del o
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (1024 * 8))()
lib.cvm_debug_stamps4(buf)
a = np.array(buf[:], dtype=np.float64).reshape(1024, 8)
a = a[a[:, 0] > 0]
names = ["row numbers", "first k-steps + weights", "k-loop", "finish (issue)", "stores acknowledged"]
for kind, label in ((0, "off-diagonal"), (1, "diagonal")):
    s = a[a[:, 6] == kind]
    if not len(s):
        continue
    d = np.diff(s[:, :6], axis=1)
    print(f"{label}: {len(s)} waves sampled, rows {int(s[0, 7])}; total {np.median(s[:, 5] - s[:, 0]):.0f} cycles (median)")
    for i, nm in enumerate(names):
        print(f"   {nm:28s} median {np.median(d[:, i]):9.0f}   mean {d[:, i].mean():9.0f}   p90 {np.percentile(d[:, i], 90):9.0f}")
