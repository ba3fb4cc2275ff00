"""Does the ROW ORDER matter to the Gram kernel?  The same 100000 x 512 problem as one segment of
all rows: the fit stage (rows in storage order, no index array) and the fold stage over one "fold"
that holds every row -- in storage order, strided (the benchmark's folds: r = f + P i), randomly
permuted, and in 16-row blocks rotated per split."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
N, K, M = 100000, 512, 16
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.rand((N, K), dtype=torch.float64, device=dev, generator=g)
Y = torch.rand((N, M), dtype=torch.float64, device=dev, generator=g)
m = CVMatrix(False, False, False, False, copy=False, device=dev, lazy_fit=False)
m.fit(X, Y)

def timed(fn, kind, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    lib.cvm_timing_enable(1)
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    a, b, na, nb = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
    lib.cvm_timing_read(C.byref(a), C.byref(na), C.byref(b), C.byref(nb))
    lib.cvm_timing_enable(0)
    return (a.value / na.value) if kind == "fit" else (b.value / nb.value)

print("fit stage (storage order, no idx): %.4f ms" % timed(lambda: m.fit(X, Y), "fit"))
rng = np.random.default_rng(0)
orders = {
    "storage order": np.arange(N),
    "strided by 10 (fold-major)": np.concatenate([np.arange(f, N, 10) for f in range(10)]),
    "strided by 7": np.concatenate([np.arange(f, N, 7) for f in range(7)]),
    "random permutation": rng.permutation(N),
    "16-row blocks shuffled": (rng.permutation(N // 16)[:, None] * 16 + np.arange(16)[None, :]).reshape(-1),
}
for name, idx in orders.items():
    b = m.prepare_folds([idx.astype(np.int64)])
    t = timed(lambda: m.training_XTX_XTY_batched(b), "fold")
    print("one fold of all rows, %-28s: %.4f ms" % (name, t))
