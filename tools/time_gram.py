"""Diagnostic: average Gram-kernel launch time (fit stage, fold stage) of a given build of
libcvmhip.so at the C3 shape, with a correctness check of the fit against float64 NumPy.
usage: python tools/time_gram.py [path/to/lib.so ...]"""
import ctypes as C, os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(path):
    sys.path.insert(0, ROOT)
    import torch
    import cvmatrix_amd._lib as L
    if path:
        L.LIB_PATH = path
    from cvmatrix_amd import CVMatrix, Partitioner
    rng = np.random.default_rng(42)
    N, K, M, P = 100000, 512, 16, 10
    X, Y, w = rng.random((N, K)), rng.random((N, M)), rng.random(N)
    m = CVMatrix(lazy_fit=False); m.fit(X, Y, w)
    b = m.prepare_folds(Partitioner(np.arange(N) % P))
    Xd, Yd, wd = m.X, m.Y, m.weights if hasattr(m, "weights") else None
    for _ in range(3):
        m.fit(X, Y, w); m.training_XTX_XTY_batched(b)
    lib = L.load()
    lib.cvm_timing_enable(1)
    for _ in range(10):
        m.fit(X, Y, w); o = m.training_XTX_XTY_batched(b)
    torch.cuda.synchronize()
    a, bb, na, nb = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
    lib.cvm_timing_read(C.byref(a), C.byref(na), C.byref(bb), C.byref(nb))
    G = m.XTX.cpu().numpy()
    Gr = (X * w[:, None]).T @ X
    err = np.abs(G - Gr).max() / np.abs(Gr).max()
    print(f"{os.path.basename(path or 'default'):28s} fit gram {a.value / na.value:.4f} ms  fold gram {bb.value / nb.value:.4f} ms  fit err {err:.1e}")


if __name__ == "__main__":
    if len(sys.argv) == 2 and sys.argv[1].startswith("--one="):
        one(sys.argv[1][6:])
    else:
        for p in (sys.argv[1:] or [""]):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one=" + p], timeout=300)
