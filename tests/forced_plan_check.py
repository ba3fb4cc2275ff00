"""Helper of test_gpu_parity.py::test_forced_split_plans (run as a subprocess: the library reads
CVM_FORCE_SPLITS once per process).  A few float64 problems through the two-stage path, the sweep
and the per-fold route under the forced row-split plan "s_off,s_diag", against the oracle."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import cvmatrix_amd as amd  # noqa: E402
from oracle.cvmatrix_oracle import OracleCVMatrix  # noqa: E402


def err(got, ref):
    got = got.double().cpu().numpy() if hasattr(got, "cpu") else np.asarray(got)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300))


def main():
    worst = 0.0
    for (N, K, M, P, seed) in ((9000, 132, 1, 3, 1), (12000, 260, 34, 4, 2), (7000, 516, 16, 2, 3), (6000, 96, 0, 5, 4)):
        rng = np.random.default_rng(seed)
        X = rng.random((N, K)) + 0.1
        Y = rng.random((N, M)) if M else None
        w = rng.random(N)
        w[rng.choice(N, 200, replace=False)] = 0
        labels = rng.integers(0, P, N)
        part = amd.Partitioner(labels)
        keys = list(part.folds_dict)
        for flags in ((True,) * 4, (False,) * 4):
            o = OracleCVMatrix(*flags)
            o.fit(X, Y, w)
            for lazy in (False, True):
                m = amd.CVMatrix(*flags, lazy_fit=lazy)
                m.fit(X, Y, w)
                if M:
                    (bx, by), _ = m.training_XTX_XTY_batched(part)
                else:
                    bx, _ = m.training_XTX_batched(part)
                worst = max(worst, err(m.XTX, o.XTX))
                for f in (0, P - 1):
                    v = part.get_validation_indices(keys[f])
                    if M:
                        (rx, ry), _ = o.training_XTX_XTY(v)
                        worst = max(worst, err(by[f], ry))
                        (cx, cy), _ = m.training_XTX_XTY(v.copy())        # (a copy: the ordinary one-fold route)
                        worst = max(worst, err(cy, ry))
                    else:
                        rx, _ = o.training_XTX(v)
                        cx, _ = m.training_XTX(v.copy())
                    worst = max(worst, err(bx[f], rx), err(cx, rx))
                    assert bool((bx[f] == bx[f].T).all())
    # mid-size folds (the fused single-split epilogue, or the two-stage route under CVM_NO_FUSED) and
    # folds of a few rows (tile kernel / whole-rows kernel, CVM_NO_DIRECT): tools/route_matrix.sh's switches
    for (N, K, M, nv, seed) in ((6000, 132, 2, 150, 5), (3000, 70, 3, 8, 6), (2400, 70, 0, 1, 7), (4000, 260, 2, 25, 8),
                              (4500, 200, 3, 90, 9)):      # (90 rows: the direct kernels in three chunks under CVM_SMALL_MAXN)
        rng = np.random.default_rng(seed)
        X = rng.random((N, K)) + 0.1
        Y = rng.random((N, M)) if M else None
        w = rng.random(N) + 0.01
        perm = rng.permutation(N)
        folds = [np.sort(perm[i:i + nv]) for i in range(0, N, nv)]
        o = OracleCVMatrix()
        o.fit(X, Y, w)
        for lazy in (False, True):
            m = amd.CVMatrix(lazy_fit=lazy)
            m.fit(X, Y, w)
            if M:
                (bx, by), _ = m.training_XTX_XTY_batched(folds)
            else:
                bx, _ = m.training_XTX_batched(folds)
            for f in (0, len(folds) // 2, len(folds) - 1):
                if M:
                    (rx, ry), _ = o.training_XTX_XTY(folds[f])
                    worst = max(worst, err(by[f], ry))
                else:
                    rx, _ = o.training_XTX(folds[f])
                worst = max(worst, err(bx[f], rx))
                assert bool((bx[f] == bx[f].T).all())
    print("plan", os.environ.get("CVM_FORCE_SPLITS"), "worst norm-wise error %.3e" % worst)
    assert worst <= 1e-10, worst


if __name__ == "__main__":
    main()
