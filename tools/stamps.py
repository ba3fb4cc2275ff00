"""Diagnostic: phase cycle shares of the Gram kernel (build with -DCVM_STAMPS)."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvmatrix_amd._lib as L
L.LIB_PATH = sys.argv[1]
from cvmatrix_amd import CVMatrix, Partitioner
rng = np.random.default_rng(42)
N, K, M, P = 100000, 512, 16, 10
X, Y, w = rng.random((N, K)), rng.random((N, M)), rng.random(N)
import os
if os.environ.get("UNWEIGHTED"): w = None
m = CVMatrix(); m.fit(X, Y, w)
b = m.prepare_folds(Partitioner(np.arange(N) % P))
for _ in range(3): m.training_XTX_XTY_batched(b)
lib = L.load(); print(lib.cvm_version().decode())
buf = (C.c_ulonglong * (1024 * 8 * 4))()
lib.cvm_debug_stamps(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8, 4).astype(np.float64)
a = a[:500]
st = np.maximum(a[:, :, 3], 1)
print("workgroups", a.shape[0], "stages/WG", a[:, 0, 3].mean())
for name, i in (("phase A", 0), ("phase B", 1), ("phase C", 2)):
    per = a[:, :, i] / st
    print(f"{name:10s} cycles/stage by wave {np.round(per.mean(0))}")
tot = (a[:, :, 0] + a[:, :, 1] + a[:, :, 2]) / st
print("total cycles/stage by wave", np.round(tot.mean(0)))
print("compute waves 0-3: A=-, B=compute, C=barrier wait; DMA loaders 4-6: A=issue, B=vmcnt wait, C=barrier wait")
