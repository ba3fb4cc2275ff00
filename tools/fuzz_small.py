"""Randomised check of the small-fold routes (tile kernel, rows kernel, inline indices, several folds
per workgroup) against the NumPy oracle: random K (aligned and not), rows per fold, fold counts,
flags, weights, dtypes.  python tools/fuzz_small.py [cases] [seed]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvmatrix_amd import CVMatrix
from oracle.cvmatrix_oracle import OracleCVMatrix

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for c in range(cases):
    dt = np.float64 if rng.random() < 0.7 else np.float32
    K = int(rng.choice([3, 17, 64, 100, 130, 200, 256, 300, 500, 512, 640, 1000]))
    if rng.random() < 0.3:
        K += int(rng.integers(1, 4))
    if os.environ.get("CVM_RESIDENT") == "1" and dt is np.float32 and rng.random() < 0.6:
        K = 1024      # (the resident route, csrc/resident.hpp: float32, K a multiple of 1024, folds of at most 32 rows, >= 4 folds)
    M = int(rng.choice([0, 1, 3, 10, 40, 70]))      # (70: two 64-response panels of XTY in the tile kernel)
    nmax = int(rng.choice([1, 2, 3, 8, 16, 32, 33, 50, 100, 128]))      # (beyond 32: several chunks, where the limit of the shape -- or CVM_SMALL_MAXN -- sends them there)
    P = int(rng.choice([1, 2, 9, 40, 300] if nmax <= 32 else [1, 2, 9, 40]))
    N = max(P * nmax + 7, 60)
    X = rng.random((N, K)).astype(dt)
    Y = rng.random((N, M)).astype(dt) if M else None
    w = rng.random(N).astype(dt) if rng.random() < 0.6 else None
    if w is not None:
        w[rng.choice(N, N // 9, replace=False)] = 0
    flags = tuple(bool(b) for b in rng.integers(0, 2, 4))
    perm = rng.permutation(N)
    folds, o = [], 0
    for f in range(P):
        n = int(rng.integers(1, nmax + 1))
        folds.append(perm[o:o + n]); o += n
    m = CVMatrix(*flags, dtype=dt, lazy_fit=bool(rng.integers(0, 2)))
    m.fit(X, Y, w)
    orc = OracleCVMatrix(*flags, dtype=np.float64)
    orc.fit(X.astype(np.float64), None if Y is None else Y.astype(np.float64), None if w is None else w.astype(np.float64))
    try:
        if M:
            (bx, by), _ = m.training_XTX_XTY_batched(folds)
        else:
            bx, _ = m.training_XTX_batched(folds); by = None
    except ValueError as e:
        # the same fold must make the oracle raise the same message
        try:
            for v in folds:
                orc.training_XTX(v)
            raise AssertionError(f"case {c}: product raised {e!r}, oracle did not")
        except ValueError as e2:
            assert str(e) == str(e2), (str(e), str(e2))
        continue
    tol = 1e-10 if dt is np.float64 else 5e-4
    for f in rng.choice(P, min(P, 4), replace=False):
        if M:
            (rx, ry), _ = orc.training_XTX_XTY(folds[f])
            e = np.abs(by[f].double().cpu().numpy() - ry).max() / max(np.abs(ry).max(), 1e-300)
            worst = max(worst, e if dt is np.float64 else 0); assert e <= tol, (c, "XTY", e, K, M, nmax, P, flags, dt)
        else:
            rx, _ = orc.training_XTX(folds[f])
        e = np.abs(bx[f].double().cpu().numpy() - rx).max() / max(np.abs(rx).max(), 1e-300)
        worst = max(worst, e if dt is np.float64 else 0); assert e <= tol, (c, "XTX", e, K, M, nmax, P, flags, dt)
        assert bool((bx[f] == bx[f].T).all())
    # one fold per call (inline indices) gives the same bits as the batch
    # (to rounding where the batch takes the rows kernel -- it sums w * (x_a * x_b), the tile kernel
    #  (w * x_a) * x_b -- and bit for bit elsewhere)
    one = m.training_XTX(folds[0])[0]
    es = np.dtype(dt).itemsize
    vw = 16 // es
    Kd = K if os.environ.get("CVM_PAD", "1") == "0" else -(-K // vw) * vw     # (the private copy's padded width)
    tc = (64 if Kd <= 64 * vw else (128 if Kd <= 128 * vw else 256)) * vw
    rows_kernel = (max(len(v) for v in folds) <= 2 and P >= 8 and Kd <= tc and 2 * Kd > tc
                   and Kd * Kd * es <= (2 << 20) + (64 << 10) and (Kd * es) % 16 == 0 and (Kd * es) % 128 != 0)
    # (a batch of folds of 8 / 16 rows or more takes mid_tile_kernel -- host.hpp: mid_default_minn -- whose sums
    #  run in the order of the Gram kernel's; the one-fold call keeps the small-fold kernels)
    nmax_rows = max(len(v) for v in folds)
    minn = int(os.environ["CVM_MID_MINN"]) if os.environ.get("CVM_MID_MINN") else (8 if (Kd < 768 or (es == 4 and Kd <= 1024)) else 16)
    mid_route = os.environ.get("CVM_MID_TILE", "1") != "0" and Kd <= 2048 and nmax_rows >= minn
    resident = (os.environ.get("CVM_RESIDENT") == "1" and dt is np.float32 and Kd % 1024 == 0 and nmax_rows <= 32 and P >= 4)
    if rows_kernel or mid_route or resident or nmax_rows > 32:      # (beyond 32 rows batch and single call may take different routes)
        assert float((one - bx[0]).abs().max()) <= (1e-12 if dt is np.float64 else 1e-5) * float(bx[0].abs().max()), (c, "per-call vs batch")
    else:
        assert torch.equal(one, bx[0]), (c, "per-call vs batch", K, M, nmax, P, dt)
print(f"{cases} cases ok, worst float64 norm-wise error {worst:.2e}")
