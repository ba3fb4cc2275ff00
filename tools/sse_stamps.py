"""Diagnostic: where workgroup 0 of pls_sse_kernel spends its cycles (build with -DCVM_STAMPS as
tools/libcvmhip_stamps.so; python tools/sse_stamps.py tools/libcvmhip_stamps.so)."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvmatrix_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
from cvmatrix_amd import CVMatrix
from cvmatrix_amd.pls import pls_fit_batched, pls_validation_sse
NAMES = ["prologue", "loop: loads issued", "loop: MFMA block", "loop: arithmetic + LDS writes", "loop: barrier", "epilogue"]
def run(N, K, M, P, A):
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    X = torch.rand((N, K), dtype=torch.float64, device="cuda", generator=g)
    Y = torch.rand((N, M), dtype=torch.float64, device="cuda", generator=g)
    m = CVMatrix(True, True, True, True, copy=False); m.fit(X, Y)
    nv = N // P
    b = m.prepare_folds([np.arange(i * nv, (i + 1) * nv) for i in range(P)])
    (XTX, XTY), st = m.training_XTX_XTY_batched(b)
    fit = pls_fit_batched(XTX, XTY, A)
    pls_validation_sse(m, b, st, fit.B)
    lib = L.load()
    buf = (C.c_ulonglong * 8)()
    lib.cvm_debug_sse_stamps(buf, 1)
    pls_validation_sse(m, b, st, fit.B)
    lib.cvm_debug_sse_stamps(buf, 1)
    a = np.array(list(buf), dtype=np.float64)
    print(f"N={N} K={K} M={M} P={P} A={A}: workgroup 0, {a[:6].sum():.0f} cycles")
    for i, nm in enumerate(NAMES):
        print(f"   {nm:32s} {a[i]:10.0f}")
run(100000, 512, 16, 10, 20)
