"""GPU (-m gpu): the device PLS (cvm_pls_fit through the C ABI) against the CPU oracle
(oracle/ikpls_oracle.py, pinned to scikit-learn) and against the scikit-learn coefficients in
tests/golden/g8_pls.npz.

Tolerance: B relative Frobenius 1e-9 against the oracle for float64 (both sides float64, but the
dominant eigenvector comes from repeated squaring here and from LAPACK there, and every later
component inherits the difference), 1e-6 against scikit-learn's NIPALS for M > 1 (its inner
iteration), 2e-3 for float32 inputs/outputs."""

import numpy as np
import pytest
import torch

from conftest import load_npz
from oracle.cvmatrix_oracle import OracleCVMatrix
from oracle.ikpls_oracle import ikpls_fit

pytestmark = pytest.mark.gpu
CASES = ["pls1_small", "pls2_small", "pls2_mid", "pls1_wide", "pls2_m16"]


@pytest.fixture(scope="module")
def pls(hip_device):
    from cvmatrix_amd import _lib
    _lib.load()
    from cvmatrix_amd import pls as mod
    return mod


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def check_against_oracle(fit, XTX, XTY, A, tol, factors=False):
    F = XTX.shape[0]
    B = fit.B.cpu().numpy()
    nf = fit.n_fit.cpu().numpy()
    for f in range(F):
        Bo, Wo, Po, Qo, Ro, n = ikpls_fit(XTX[f].astype(np.float64), XTY[f].astype(np.float64), A)
        assert nf[f] == n, (f, nf[f], n)
        for a in range(A):
            if a < n:
                assert rel(B[f, a], Bo[a]) <= tol, (f, a, rel(B[f, a], Bo[a]))
            else:
                assert np.all(B[f, a] == 0)
        if factors:
            for got, ref in ((fit.W, Wo), (fit.P, Po), (fit.R, Ro), (fit.Q, Qo)):
                got = got[f].cpu().numpy()
                for a in range(n):
                    sgn = np.sign(np.dot(got[:, a], ref[:, a])) or 1.0
                    assert rel(sgn * got[:, a], ref[:, a]) <= 10 * tol, (f, a)


@pytest.mark.parametrize("name", CASES)
def test_golden_sklearn_and_oracle(pls, name):
    g = load_npz("g8_pls.npz")
    XTX, XTY, Bref = g[f"{name}/XTX"], g[f"{name}/XTY"], g[f"{name}/B"]
    A = Bref.shape[0]
    fit = pls.pls_fit_batched(torch.from_numpy(XTX).cuda(), torch.from_numpy(XTY).cuda(), A, return_factors=True)
    check_against_oracle(fit, XTX[None], XTY[None], A, 1e-9, factors=True)
    B = fit.B.cpu().numpy()[0]
    tol = 1e-10 if XTY.shape[1] == 1 else 1e-6
    for a in range(A):
        assert rel(B[a], Bref[a]) <= tol, (a, rel(B[a], Bref[a]))


def fold_matrices(N, K, M, P, seed, dtype=np.float64, weighted=True):
    rng = np.random.default_rng(seed)
    L = rng.standard_normal((N, 5))
    X = (L @ rng.standard_normal((5, K)) + 0.5 * rng.standard_normal((N, K))).astype(dtype)
    Y = (L[:, :2] @ rng.standard_normal((2, M)) + 0.1 * rng.standard_normal((N, M))).astype(dtype)
    w = (rng.random(N) + 0.1).astype(dtype) if weighted else None
    folds = [np.arange(N)[np.arange(N) % P == f] for f in range(P)]
    return X, Y, w, folds


@pytest.mark.parametrize("N,K,M,P,A", [
    (400, 24, 3, 4, 8),        # few folds: several slices per fold, XTX slice in LDS
    (600, 64, 16, 3, 12),
    (900, 40, 1, 300, 6),      # many folds: one workgroup per fold
    (500, 33, 5, 7, 9),        # odd K: ragged last slice
    (300, 200, 2, 2, 10),
    (500, 40, 20, 4, 8),       # 16 < M <= 32: the M x M matrix in 2 x 2 MFMA blocks
    (700, 48, 32, 300, 6),
    (400, 36, 17, 2, 5),
    (500, 40, 33, 4, 6),       # 32 < M <= 64: the squarings in LDS by all waves (3 x 3, 4 x 4 blocks)
    (600, 64, 48, 3, 8),
    (700, 48, 64, 300, 5),
    (400, 72, 50, 2, 6),
    (1500, 512, 64, 10, 5),    # too large for the replicated-state kernel: XTY cut in slices
])
def test_folds_from_cvmatrix_match_oracle(pls, N, K, M, P, A):
    from cvmatrix_amd import CVMatrix
    X, Y, w, folds = fold_matrices(N, K, M, P, seed=K + M)
    m = CVMatrix(True, True, True, True, dtype=np.float64)
    m.fit(X, Y, w)
    (XTX, XTY), _ = m.training_XTX_XTY_batched(m.prepare_folds(folds))
    fit = pls.pls_fit_batched(XTX, XTY, A, return_factors=True)

    check_against_oracle(fit, XTX.cpu().numpy(), XTY.cpu().numpy(), A, 1e-9, factors=True)
    # and the whole chain against the CPU oracle of the hot path
    o = OracleCVMatrix(True, True, True, True, dtype=np.float64)
    o.fit(X, Y, w)
    (oXTX, oXTY), _ = o.training_XTX_XTY(folds[0])
    Bo, *_ = ikpls_fit(oXTX, oXTY, A)
    assert rel(fit.B[0, A - 1].cpu().numpy(), Bo[A - 1]) <= 1e-8


def test_sliced_and_unsliced_agree(pls):
    # the same fold alone (many slices) and inside a batch of 300 copies (one slice each)
    rng = np.random.default_rng(3)
    X = rng.standard_normal((500, 48)); Y = rng.standard_normal((500, 4))
    XTX = torch.from_numpy(X.T @ X).cuda(); XTY = torch.from_numpy(X.T @ Y).cuda()
    one = pls.pls_fit_batched(XTX, XTY, 10)
    many = pls.pls_fit_batched(XTX.expand(300, -1, -1).contiguous(), XTY.expand(300, -1, -1).contiguous(), 10)
    a, b = one.B[0].cpu().numpy(), many.B.cpu().numpy()
    assert np.all(b == b[0:1])                      # every copy bitwise the same
    assert rel(a, b[0]) <= 1e-11                    # slicing changes summation order only


def test_float32(pls):
    from cvmatrix_amd import CVMatrix
    X, Y, w, folds = fold_matrices(800, 32, 4, 5, seed=9, dtype=np.float32)
    m = CVMatrix(True, True, True, True, dtype=np.float32)
    m.fit(X, Y, w)
    (XTX, XTY), _ = m.training_XTX_XTY_batched(m.prepare_folds(folds))
    fit = pls.pls_fit_batched(XTX, XTY, 6)
    assert fit.B.dtype == torch.float32
    B = fit.B.cpu().numpy()
    for f in range(5):
        Bo, *_ = ikpls_fit(XTX[f].cpu().numpy().astype(np.float64), XTY[f].cpu().numpy().astype(np.float64), 6)
        for a in range(6):
            assert rel(B[f, a], Bo[a]) <= 2e-3


def test_stops_when_xty_is_exhausted(pls):
    XTX = torch.eye(6, dtype=torch.float64, device="cuda").repeat(3, 1, 1)
    XTY = torch.zeros((3, 6, 1), dtype=torch.float64, device="cuda")
    XTY[:, 0, 0] = 3.0
    fit = pls.pls_fit_batched(XTX, XTY, 4, return_factors=True)
    assert fit.n_fit.tolist() == [1, 1, 1]
    B = fit.B.cpu().numpy()
    assert np.all(B[:, 1:] == 0) and np.all(B[:, 0, 0, 0] == 3.0)


def test_wide_k_streams_xtx(pls):
    # K large enough that a slice of XTX does not fit in LDS (streamed from HBM per component)
    rng = np.random.default_rng(21)
    K = 1536
    X = rng.standard_normal((400, K)); Y = rng.standard_normal((400, 1))
    XTX = (X.T @ X); XTY = (X.T @ Y)
    assert not pls.pls_plan(2, K, 1, 5)["xtx_in_lds"]
    fit = pls.pls_fit_batched(torch.from_numpy(np.stack([XTX, XTX])).cuda(),
                              torch.from_numpy(np.stack([XTY, 2 * XTY])).cuda(), 5)
    check_against_oracle(fit, np.stack([XTX, XTX]), np.stack([XTY, 2 * XTY]), 5, 1e-9)


def test_argument_errors(pls):
    XTX = torch.eye(4, dtype=torch.float64, device="cuda")[None]
    XTY = torch.ones((1, 4, 2), dtype=torch.float64, device="cuda")
    with pytest.raises(ValueError):
        pls.pls_fit_batched(XTX, XTY, 0)
    with pytest.raises(ValueError):
        pls.pls_fit_batched(XTX, XTY.float(), 2)
    with pytest.raises(ValueError):
        pls.pls_fit_batched(XTX, torch.ones((1, 4, 65), dtype=torch.float64, device="cuda"), 2)
    with pytest.raises(TypeError):
        pls.pls_fit_batched(XTX.cpu(), XTY.cpu(), 2)


def test_end_to_end_equals_refitting_each_fold_with_sklearn(pls):
    """CVMatrix (centre + scale, ddof=1, unweighted) -> device PLS -> predictions on the validation
    rows, against scikit-learn refitted from scratch on every training set (PLS1: no inner
    iteration, agreement to rounding)."""
    cross = pytest.importorskip("sklearn.cross_decomposition")
    from cvmatrix_amd import CVMatrix
    rng = np.random.default_rng(17)
    N, K, P, A = 240, 12, 4, 5
    L = rng.standard_normal((N, 4))
    X = L @ rng.standard_normal((4, K)) + 0.3 * rng.standard_normal((N, K))
    Y = L[:, :2] @ rng.standard_normal((2, 1)) + 0.05 * rng.standard_normal((N, 1))
    folds = [np.arange(N)[np.arange(N) % P == f] for f in range(P)]
    m = CVMatrix(True, True, True, True, ddof=1, dtype=np.float64)
    m.fit(X, Y)
    (XTX, XTY), (muX, sdX, muY, sdY) = m.training_XTX_XTY_batched(m.prepare_folds(folds))
    fit = pls.pls_fit_batched(XTX, XTY, A)
    B = fit.B.cpu().numpy()
    muX, sdX, muY, sdY = (t.cpu().numpy() for t in (muX, sdX, muY, sdY))
    for f, val in enumerate(folds):
        train = np.setdiff1d(np.arange(N), val)
        Xs = (X[val] - muX[f]) / sdX[f]
        for a in range(A):
            ref = cross.PLSRegression(n_components=a + 1, scale=True).fit(X[train], Y[train]).predict(X[val])
            got = Xs @ B[f, a] * sdY[f] + muY[f]
            assert np.abs(got - ref.reshape(got.shape)).max() <= 1e-9 * np.abs(ref).max(), (f, a)


def test_sliced_runs_are_bitwise_repeatable(pls):
    # the per-fold barrier and the exchange between slices: same bits run after run
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    for F, K, M, A in ((4, 768, 8, 8), (2, 1024, 1, 6)):
        X = torch.randn((F, 2 * K, K), dtype=torch.float64, device="cuda", generator=g)
        Y = torch.randn((F, 2 * K, M), dtype=torch.float64, device="cuda", generator=g)
        XTX, XTY = X.transpose(1, 2) @ X, X.transpose(1, 2) @ Y
        assert pls.pls_plan(F, K, M, A)["slices"] > 1
        ref = pls.pls_fit_batched(XTX, XTY, A, return_factors=True)
        for _ in range(25):
            out = pls.pls_fit_batched(XTX, XTY, A, return_factors=True)
            for a, b in zip((out.B, out.W, out.P, out.Q, out.R), (ref.B, ref.W, ref.P, ref.Q, ref.R)):
                assert torch.equal(a, b)


def test_example_fast_cv_matches_refits(pls):
    """examples/fast_cv_pls.py: cross-validated RMSE per number of components equals the one from
    scikit-learn models refitted on every training set (PLS1)."""
    cross = pytest.importorskip("sklearn.cross_decomposition")
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fast_cv_pls", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "fast_cv_pls.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    rng = np.random.default_rng(2)
    N, K, P, A = 300, 10, 5, 4
    L = rng.standard_normal((N, 3))
    X = L @ rng.standard_normal((3, K)) + 0.3 * rng.standard_normal((N, K))
    Y = L[:, :1] * 2.0 + 0.1 * rng.standard_normal((N, 1))
    labels = np.arange(N) % P
    got = ex.fast_cv_rmse(X, Y, labels, A)
    sse = np.zeros(A)
    for f in range(P):
        val, tr = labels == f, labels != f
        for a in range(A):
            pred = cross.PLSRegression(n_components=a + 1, scale=True).fit(X[tr], Y[tr]).predict(X[val])
            sse[a] += ((pred.reshape(-1) - Y[val].reshape(-1)) ** 2).sum()
    np.testing.assert_allclose(got[:, 0], np.sqrt(sse / N), rtol=1e-9)


@pytest.mark.parametrize("K,M,A,F", [(3, 2, 3, 4), (1, 1, 1, 3), (5, 4, 5, 300), (7, 1, 7, 2), (2, 3, 2, 1)])
def test_tiny_shapes(pls, K, M, A, F):
    rng = np.random.default_rng(K * 10 + M)
    XTX = np.stack([(lambda Z: Z.T @ Z)(rng.standard_normal((4 * K + 3, K))) for _ in range(F)])
    XTY = np.stack([rng.standard_normal((K, M)) for _ in range(F)])
    fit = pls.pls_fit_batched(torch.from_numpy(XTX).cuda(), torch.from_numpy(XTY).cuda(), A, return_factors=True)
    check_against_oracle(fit, XTX, XTY, A, 1e-8)


def test_four_barrier_kernel_still_agrees(pls):
    """CVM_PLS_NO_REP=1 (read once per process): the row-sliced kernel with four barriers per component,
    which problems with 65-255 folds and wide response blocks still take, over this file's oracle and
    repeatability tests."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, CVM_PLS_NO_REP="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x",
                        "-k", "not four_barrier and not example"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("dtype,N,K,M,P,A,weighted,flags", [
    (np.float64, 3000, 64, 4, 6, 7, True, (True, True, True, True)),
    (np.float64, 2111, 130, 2, 5, 5, False, (True, True, False, False)),
    (np.float64, 1500, 36, 16, 3, 20, True, (False, False, False, False)),
    (np.float32, 4000, 128, 3, 8, 6, True, (True, True, True, True)),
    (np.float64, 2000, 96, 40, 4, 6, True, (True, True, True, True)),      # 32 < M <= 64
    (np.float64, 1200, 48, 40, 3, 12, True, (True, True, True, True)),     # 480 columns: two column groups
    (np.float32, 1500, 64, 8, 4, 30, False, (True, True, True, True)),     # float32, 16-byte loads, 240 columns
    (np.float64, 900, 35, 3, 3, 4, True, (True, False, True, False)),      # odd K, odd M: the scalar loads
])
def test_validation_sse_on_the_device(pls, dtype, N, K, M, P, A, weighted, flags):
    """cvm_pls_validation_sse: the squared validation errors of every fold's models (every number of
    components) against the same quantity computed with plain torch operations from the same
    coefficients -- ragged folds, with and without weights / centring / scaling."""
    import cvmatrix_amd as amd
    from cvmatrix_amd.pls import cv_rmse, pls_fit_batched, pls_validation_sse

    rng = np.random.default_rng(N + K)
    L = rng.standard_normal((N, 5))
    X = (L @ rng.standard_normal((5, K)) + 0.3 * rng.standard_normal((N, K))).astype(dtype)
    Y = (L[:, :3] @ rng.standard_normal((3, M)) + 0.1 * rng.standard_normal((N, M))).astype(dtype)
    w = (rng.random(N) + 0.1).astype(dtype) if weighted else None
    labels = rng.integers(0, P, N)
    p = amd.Partitioner(labels)
    if K % 2 or M % 2:
        # odd shapes are padded in a private copy (and refused here); on the caller's own device
        # arrays they stay as they are: the kernel's scalar loads
        cvm = amd.CVMatrix(*flags, dtype=dtype, copy=False)
        cvm.fit(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda(), None if w is None else torch.from_numpy(w).cuda())
    else:
        cvm = amd.CVMatrix(*flags, dtype=dtype)
        cvm.fit(X, Y, w)
    batch = cvm.prepare_folds(p)
    (XTX, XTY), stats = cvm.training_XTX_XTY_batched(batch)
    B = pls_fit_batched(XTX, XTY, A).B
    sse, wsum = pls_validation_sse(cvm, batch, stats, B)
    muX, sdX, muY, sdY = stats
    f64 = torch.float64
    for f, key in enumerate(p.folds_dict):
        val = torch.from_numpy(p.get_validation_indices(key)).to(B.device)
        Xs = cvm.X[val].to(f64)
        if muX is not None: Xs = Xs - muX[f].to(f64)
        if sdX is not None: Xs = Xs / sdX[f].to(f64)
        pred = torch.matmul(Xs, B[f].to(f64))
        if sdY is not None: pred = pred * sdY[f].to(f64)
        if muY is not None: pred = pred + muY[f].to(f64)
        e2 = (pred - cvm.Y[val].to(f64)) ** 2
        wv = cvm.weights[val].to(f64) if weighted else torch.ones((val.numel(), 1), dtype=f64, device=B.device)
        ref = (e2 * wv).sum(dim=1)
        tol = 1e-10 if dtype is np.float64 else 2e-4
        assert float((sse[f] - ref).abs().max()) <= tol * float(ref.abs().max()), (f, float((sse[f] - ref).abs().max()))
        assert abs(float(wsum[f]) - float(wv.sum())) <= 1e-12 * float(wv.sum())
    r = cv_rmse(sse, wsum)
    assert r.shape == (A, M) and bool(torch.isfinite(r).all())


def test_randomised_shapes_against_the_oracle():
    """tools/fuzz_pls.py: random N, K, M (1 ... 64), fold counts, components, element types, flags and
    weights -- the coefficients of every well-determined component against the NumPy oracle
    (float64 1e-8) and the validation errors against the same formula in torch operations (1e-9)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_pls.py"), "120", "21"], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "cases ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.parametrize("env", [{}, {"CVM_PLS_NO_REP": "1"}])
def test_a_timed_out_sliced_launch_is_recomputed_not_poisoned(env):
    """The routes that cut a fold into slices spin on each other; a slice that finds no place to run (another
    stream's kernel, a masked CU) must cost time, not results.  ``CVM_PLS_TEST_TIMEOUT`` launches them one
    workgroup short with a short spin limit: the barrier of one fold times out, the library recomputes every
    fold with the one-workgroup-per-fold kernel in the same call (status 2) and the coefficients are the
    oracle's."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import numpy as np, torch, sys
sys.path.insert(0, %r)
from cvmatrix_amd import CVMatrix, Partitioner
from cvmatrix_amd.pls import pls_fit_batched, pls_plan
from oracle.ikpls_oracle import ikpls_fit
rng = np.random.default_rng(5)
N, K, M, P, A = 3000, 512, 4, 10, 6
X, Y = rng.random((N, K)), rng.random((N, M))
m = CVMatrix(); m.fit(X, Y)
(xtx, xty), _ = m.training_XTX_XTY_batched(Partitioner(np.arange(N) %% P))
assert pls_plan(P, K, M, A)["slices"] > 1, pls_plan(P, K, M, A)
fit = pls_fit_batched(xtx, xty, A)
assert pls_fit_batched.last_status == 2, pls_fit_batched.last_status
for f in (0, 4, 9):
    Bo, *_ = ikpls_fit(xtx[f].cpu().numpy(), xty[f].cpu().numpy(), A)
    err = np.linalg.norm(fit.B[f].cpu().numpy() - Bo) / np.linalg.norm(Bo)
    assert err <= 1e-9, (f, err)
assert int(fit.n_fit.min()) == A
print("recovered ok")
""" % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CVM_PLS_TEST_TIMEOUT="1", **env), cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "recovered ok" in r.stdout

