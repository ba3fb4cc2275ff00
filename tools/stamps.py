"""Diagnostic: phase cycle shares of the Gram kernel (build with -DCVM_STAMPS)."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvmatrix_amd._lib as L
L.LIB_PATH = sys.argv[1]
from cvmatrix_amd import CVMatrix, Partitioner
rng = np.random.default_rng(42)
N, K, M, P = 100000, 512, 16, 10
X, Y, w = rng.random((N, K)), rng.random((N, M)), rng.random(N)
m = CVMatrix(); m.fit(X, Y, w)
b = m.prepare_folds(Partitioner(np.arange(N) % P))
for _ in range(3): m.training_XTX_XTY_batched(b)
lib = L.load(); print(lib.cvm_version().decode())
buf = (C.c_ulonglong * (1024 * 8 * 4))()
lib.cvm_debug_stamps(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8, 4).astype(np.float64)
a = a[a[:, 0, 3] > 0]; nw = 4 if a[:, 4:, 3].sum() == 0 else 8; a = a[:, :nw]
st = a[:, :, 3]
print("workgroups", a.shape[0], "stages/WG", st.mean())
for name, i in (("issue(top)", 0), ("compute", 1), ("tail(write+barrier)", 2)):
    per = a[:, :, i] / st
    print(f"{name:22s} cycles/stage: mean {per.mean():8.0f}  by wave {np.round(per.mean(0))}")
tot = (a[:, :, 0] + a[:, :, 1] + a[:, :, 2]) / st
print("total cycles/stage", tot.mean(), " (MFMA-bound ideal 4096/CU-stage with 2 waves/SIMD)")
