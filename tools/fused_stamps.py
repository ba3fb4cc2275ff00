"""Diagnostic: where an off-diagonal workgroup of the fused route (one unit per fold) spends its cycles
(build with -DCVM_STAMPS): python tools/fused_stamps.py tools/libcvmhip_stamps.so <folds>"""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvmatrix_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
from cvmatrix_amd import CVMatrix
rng = np.random.default_rng(42)
N, K, M, P = 100000, 512, 16, int(sys.argv[2])
X, Y, w = rng.random((N, K)), rng.random((N, M)), rng.random(N)
m = CVMatrix(lazy_fit=False); m.fit(X, Y, w)
nv = N // P
b = m.prepare_folds([np.arange(f, N, P)[:nv] for f in range(P)])
for _ in range(3): m.training_XTX_XTY_batched(b)
torch.cuda.synchronize()
lib = L.load()
buf = (C.c_ulonglong * (1024 * 8))()
lib.cvm_debug_stamps4(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.float64)
a = a[a[:, 7] > 0]
names = ["prologue", "stage loop", "f0->f1 (drain barrier, rs, dump)", "B_dump wait", "direct", "B_parked wait", "mirror+stores drain", "total"]
print("off-diagonal workgroups seen", len(a))
for i, nm in enumerate(names): print(f"  {nm:36s} {a[:, i].mean():9.0f} cycles")
