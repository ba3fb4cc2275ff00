"""Fast cross-validation of a PLS model, end to end on one MI355X -- what the reference's README
describes as the use of its training matrices (README.md:23: "fast cross-validation algorithms
combined with Improved Kernel PLS", the out-of-tree `ikpls` package).

    python examples/fast_cv_pls.py [N K M folds components]

1. CVMatrix.fit + training_XTX_XTY_batched   training-set XtX, XtY, means, stds of every fold   (HIP)
2. pls_fit_batched                           A-component PLS coefficients of every fold          (HIP)
3. pls_validation_sse                        squared validation errors of every fold's models        (HIP)
   -> RMSE per number of components.  (Shapes whose device copies are padded -- odd K, float64 with
   odd M -- take the same formula in plain torch operations.)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvmatrix_amd import CVMatrix, Partitioner  # noqa: E402
from cvmatrix_amd.pls import cv_rmse, pls_fit_batched, pls_validation_sse  # noqa: E402


def fast_cv_rmse(X, Y, labels, A, weights=None):
    """RMSE[a, m] over all validation rows for PLS models with a+1 components (a < A), every row
    predicted by the model that was trained without its fold.  X, Y: NumPy arrays."""
    p = Partitioner(labels)
    cvm = CVMatrix(center_X=True, center_Y=True, scale_X=True, scale_Y=True, ddof=1, dtype=np.float64)
    cvm.fit(X, Y, weights)
    batch = cvm.prepare_folds(p)
    (XTX, XTY), (muX, sdX, muY, sdY) = cvm.training_XTX_XTY_batched(batch)
    B = pls_fit_batched(XTX, XTY, A).B                                   # (F, A, K, M)
    if cvm._Kd == cvm._Ku and (cvm._Md or 0) == (cvm._Mu or 0):
        sse_f, wsum_f = pls_validation_sse(cvm, batch, (muX, sdX, muY, sdY), B)
        return cv_rmse(sse_f, wsum_f).cpu().numpy()
    sse = torch.zeros((A, Y.shape[1]), dtype=torch.float64, device=B.device)
    wsum = 0.0
    for f, key in enumerate(p.folds_dict):
        val = torch.from_numpy(p.get_validation_indices(key)).to(B.device)
        Xs = (cvm.X[val] - muX[f]) / sdX[f]                              # training-set centring/scaling
        pred = torch.matmul(Xs, B[f]) * sdY[f] + muY[f]                  # (A, n_val, M)
        err2 = (pred - cvm.Y[val]) ** 2
        if weights is not None:
            wv = cvm.weights[val]
            err2 = err2 * wv
            wsum += float(wv.sum())
        else:
            wsum += float(val.numel())
        sse += err2.sum(dim=1)
    return torch.sqrt(sse / wsum).cpu().numpy()


def main():
    N, K, M, P, A = (int(a) for a in sys.argv[1:6]) if len(sys.argv) >= 6 else (20000, 128, 2, 10, 12)
    rng = np.random.default_rng(0)
    L = rng.standard_normal((N, 6))
    X = L @ rng.standard_normal((6, K)) + 0.2 * rng.standard_normal((N, K))
    Y = L[:, :3] @ rng.standard_normal((3, M)) + 0.1 * rng.standard_normal((N, M))
    rmse = fast_cv_rmse(X, Y, np.arange(N) % P, A)
    print("components  RMSE per response (10-fold cross-validation)")
    for a in range(A):
        print(f"{a + 1:10d}  " + "  ".join(f"{v:.5f}" for v in rmse[a]))
    best = int(np.argmin(rmse.mean(axis=1))) + 1
    print(f"lowest mean RMSE with {best} components")
    return rmse


if __name__ == "__main__":
    main()
