"""Race screen for the sliced PLS kernel: the per-fold barrier and the device-coherent exchange
must give bitwise identical results run after run, also while another stream keeps the memory
system busy.  python tools/soak_pls.py [repeats]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd.pls import pls_fit_batched, pls_plan

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = torch.Generator(device="cuda"); g.manual_seed(0)
bad = 0
for (F, K, M, A, dt) in [(10, 512, 16, 20, torch.float64), (3, 1536, 1, 8, torch.float64), (20, 2048, 1, 6, torch.float32),
                         (64, 1024, 32, 6, torch.float64), (7, 200, 5, 12, torch.float64)]:
    X = torch.randn((F, 2 * K, K), dtype=dt, device="cuda", generator=g)
    Y = torch.randn((F, 2 * K, M), dtype=dt, device="cuda", generator=g)
    XTX = X.transpose(1, 2) @ X; XTY = X.transpose(1, 2) @ Y
    del X, Y
    ref = pls_fit_batched(XTX, XTY, A, return_factors=True)
    noise = torch.empty(256 * 1024 * 1024 // 8, dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()
    n_bad = 0
    for i in range(reps):
        if i % 2:
            with torch.cuda.stream(side):
                noise.add_(1.0)                      # HBM traffic next to the kernel
        out = pls_fit_batched(XTX, XTY, A, return_factors=True, check=False)
        same = all(torch.equal(a, b) for a, b in zip((out.B, out.W, out.P, out.Q, out.R), (ref.B, ref.W, ref.P, ref.Q, ref.R)))
        n_bad += 0 if same else 1
    torch.cuda.synchronize()
    print(f"F={F} K={K} M={M} A={A} {str(dt)[6:]} plan={pls_plan(F, K, M, A, np.float64 if dt == torch.float64 else np.float32)}: "
          f"{reps} runs, {n_bad} differ")
    bad += n_bad
    del noise
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
