"""CPU: the PLS oracle (oracle/ikpls_oracle.py) against its pins -- scikit-learn's NIPALS
coefficients stored in tests/golden/g8_pls.npz (tests/golden/make_golden_pls.py) and, when
scikit-learn is importable, the same fit run live -- and the host-side slicing plan of the
device PLS."""

import numpy as np
import pytest

from conftest import load_npz
from oracle.ikpls_oracle import ikpls_fit

CASES = ["pls1_small", "pls2_small", "pls2_mid", "pls1_wide", "pls2_m16"]
# PLS1 has no inner iteration in NIPALS: agreement to rounding; PLS2 is limited by the
# convergence of scikit-learn's power iteration (tol 1e-15 on the weight update)
TOL = {"pls1_small": 1e-11, "pls1_wide": 1e-10, "pls2_small": 1e-6, "pls2_mid": 1e-6, "pls2_m16": 1e-6}


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_sklearn_golden(name):
    g = load_npz("g8_pls.npz")
    Bref = g[f"{name}/B"]
    B, W, P, Q, R, n_fit = ikpls_fit(g[f"{name}/XTX"], g[f"{name}/XTY"], Bref.shape[0])
    assert n_fit == Bref.shape[0]
    for a in range(Bref.shape[0]):
        err = np.linalg.norm(B[a] - Bref[a]) / np.linalg.norm(Bref[a])
        assert err <= TOL[name], (name, a, err)


def test_oracle_matches_sklearn_live():
    cross = pytest.importorskip("sklearn.cross_decomposition")
    rng = np.random.default_rng(11)
    X = rng.standard_normal((120, 10)) @ rng.standard_normal((10, 10))
    Y = X[:, :3] @ rng.standard_normal((3, 1)) + 0.05 * rng.standard_normal((120, 1))
    Xc, Yc = X - X.mean(0), Y - Y.mean(0)
    B, *_ = ikpls_fit(Xc.T @ Xc, Xc.T @ Yc, 6)
    for a in range(6):
        m = cross.PLSRegression(n_components=a + 1, scale=False).fit(Xc, Yc)
        ref = np.asarray(m.coef_).T.reshape(10, 1)
        assert np.linalg.norm(B[a] - ref) <= 1e-10 * np.linalg.norm(ref)


def test_oracle_properties():
    rng = np.random.default_rng(5)
    X = rng.standard_normal((80, 12))
    Y = rng.standard_normal((80, 4))
    XTX, XTY = X.T @ X, X.T @ Y
    B, W, P, Q, R, n_fit = ikpls_fit(XTX, XTY, 12)
    assert n_fit == 12
    # scores T = X R are mutually orthogonal; R^T P = I; with all K components PLS = least squares
    T = X @ R
    TT = T.T @ T
    assert np.abs(TT - np.diag(np.diag(TT))).max() <= 1e-9 * np.abs(TT).max()
    assert np.abs(R.T @ P - np.eye(12)).max() <= 1e-9
    ols = np.linalg.solve(XTX, XTY)
    assert np.linalg.norm(B[-1] - ols) <= 1e-8 * np.linalg.norm(ols)
    assert np.allclose(np.linalg.norm(W, axis=0), 1.0)


def test_oracle_stops_when_xty_is_exhausted():
    XTX = np.eye(6)
    XTY = np.zeros((6, 1)); XTY[0, 0] = 3.0
    B, W, P, Q, R, n_fit = ikpls_fit(XTX, XTY, 4)
    assert n_fit == 1
    assert np.all(B[1:] == 0) and B[0][0, 0] == 3.0


def test_device_pls_plan_is_host_logic():
    from cvmatrix_amd.pls import pls_plan
    p = pls_plan(10, 512, 16, 20)                       # few folds: the one-barrier kernel, a fold's slices on one XCD
    assert p["slices"] * p["rows"] >= 512 and p["slices"] * 2 <= 32 and "one barrier" in p["kernel"]
    assert p["lds_bytes"] <= 150 * 1024 and p["folds_per_launch"] >= 10
    p = pls_plan(64, 1024, 32, 20)                      # the whole deflated XTY does not fit in LDS: row slices, four barriers
    assert p["slices"] > 1 and p["slices"] * p["folds_per_launch"] <= 256 and "four barriers" in p["kernel"]
    p = pls_plan(1000, 500, 10, 30)                     # many folds: one workgroup per fold
    assert p["slices"] == 1 and p["rows"] == 500 and not p["xtx_in_lds"]
    p = pls_plan(20, 4096, 1, 20, np.float32)           # wide K: sliced, streamed
    assert p["slices"] > 1 and p["slices"] * p["folds_per_launch"] <= 256
    with pytest.raises(RuntimeError):
        pls_plan(4, 16, 65, 2)
