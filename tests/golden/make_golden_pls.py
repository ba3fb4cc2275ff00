"""Generates tests/golden/g8_pls.npz: PLS regression coefficients from scikit-learn's NIPALS
(``PLSRegression(scale=False)``, inner loop converged to 1e-15) for seeded inputs, the pin of
``oracle/ikpls_oracle.py``.  Run in the build container:  python tests/golden/make_golden_pls.py

The consumer the reference names (``ikpls``) is neither vendored nor installed, so its
algorithm is pinned through the model it defines, which scikit-learn 1.7.2 also fits.
"""
import os
import numpy as np
import sklearn
from sklearn.cross_decomposition import PLSRegression

CASES = [  # name, N, K, M, A, seed
    ("pls1_small", 60, 8, 1, 5, 0),
    ("pls2_small", 60, 8, 3, 6, 1),
    ("pls2_mid", 300, 40, 5, 12, 2),
    ("pls1_wide", 80, 50, 1, 10, 3),
    ("pls2_m16", 500, 64, 16, 20, 4),
]


def make_case(N, K, M, seed):
    rng = np.random.default_rng(seed)
    L = rng.standard_normal((N, 6))
    X = L @ rng.standard_normal((6, K)) + 0.3 * rng.standard_normal((N, K))
    Y = L[:, :3] @ rng.standard_normal((3, M)) + X[:, : min(K, 4)] @ rng.standard_normal((min(K, 4), M)) \
        + 0.1 * rng.standard_normal((N, M))
    return X, Y


def main():
    out = {"sklearn_version": np.array(sklearn.__version__)}
    for name, N, K, M, A, seed in CASES:
        X, Y = make_case(N, K, M, seed)
        Xc, Yc = X - X.mean(0), Y - Y.mean(0)
        out[f"{name}/XTX"] = Xc.T @ Xc
        out[f"{name}/XTY"] = Xc.T @ Yc
        coefs = []
        for a in range(1, A + 1):
            m = PLSRegression(n_components=a, scale=False, tol=1e-15, max_iter=100000).fit(Xc, Yc)
            coefs.append(np.asarray(m.coef_).T.reshape(K, M))   # sklearn >= 1.3: coef_ is (M, K)
        out[f"{name}/B"] = np.stack(coefs)
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "g8_pls.npz"), **out)
    print("wrote g8_pls.npz:", [c[0] for c in CASES])


if __name__ == "__main__":
    main()
