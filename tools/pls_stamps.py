"""Diagnostic: where workgroup 0 of the PLS kernel spends its cycles (build with -DCVM_STAMPS:
hipcc ... -DCVM_STAMPS -o tools/libcvmhip_stamps.so; python tools/pls_stamps.py tools/libcvmhip_stamps.so)."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvmatrix_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
from cvmatrix_amd.pls import pls_fit_batched, pls_plan
NAMES = ["prologue", "1 partial S", "barrier 1", "2a eig", "2b w, P^T w", "barrier 2", "3 r", "barrier 3",
         "4 matvec", "4 partial tt, v", "barrier 4", "5 deflate, B"]
def run(F, K, M, A, dtype=torch.float64):
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    X = torch.randn((F, 2 * K, K), dtype=dtype, device="cuda", generator=g)
    Y = torch.randn((F, 2 * K, M), dtype=dtype, device="cuda", generator=g)
    XTX = X.transpose(1, 2) @ X; XTY = X.transpose(1, 2) @ Y
    pls_fit_batched(XTX, XTY, A)
    lib = L.load()
    buf = (C.c_ulonglong * 16)()
    lib.cvm_debug_pls_stamps(buf, 1)
    pls_fit_batched(XTX, XTY, A)
    lib.cvm_debug_pls_stamps(buf, 1)
    a = np.array(list(buf), dtype=np.float64)
    print(f"F={F} K={K} M={M} A={A} {pls_plan(F, K, M, A)}  total {a[:12].sum()/A:.0f} shader-clock cycles/component")
    print(f"   squarings          {a[15] / A:10.1f} per component")
    print(f"   within 2a, cumulative: xreduce {a[14]/A:.0f}, squarings done {a[12]/A:.0f}, column picked {a[13]/A:.0f} cycles/component")
    a[12:16] = 0
    for i, nm in enumerate(NAMES):
        print(f"   {nm:18s} {a[i]/ (1 if i == 0 else A):10.0f} cycles{'' if i == 0 else '/component'}")
if len(sys.argv) > 2:
    for spec in sys.argv[2:]:
        run(*[int(v) for v in spec.split(",")])
    sys.exit(0)
run(10, 512, 16, 20)
run(1000, 512, 16, 20)
run(20, 4096, 1, 10, torch.float32)
run(64, 1024, 32, 20)
run(100, 512, 16, 20)
