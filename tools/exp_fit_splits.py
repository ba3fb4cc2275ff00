import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cvmatrix_amd import CVMatrix, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
N, K, M = 100000, 512, 16
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.rand((N, K), dtype=torch.float64, device=dev, generator=g)
Y = torch.rand((N, M), dtype=torch.float64, device=dev, generator=g)
w = torch.rand((N,), dtype=torch.float64, device=dev, generator=g)
m = CVMatrix(copy=False, device=dev, lazy_fit=False)
for _ in range(300): m.fit(X, Y, w)
torch.cuda.synchronize()
lib.cvm_timing_enable(1)
for _ in range(200): m.fit(X, Y, w)
torch.cuda.synchronize()
a, b, na, nb = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
lib.cvm_timing_read(C.byref(a), C.byref(na), C.byref(b), C.byref(nb))
print("fit Gram, splits", os.environ.get("CVM_FORCE_SPLITS"), ": %.4f ms" % (a.value / na.value))
