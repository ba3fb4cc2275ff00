"""CPU: the oracle (oracle/cvmatrix_oracle.py) against the reference's golden vectors.

This is the pin that lets the GPU parity tests trust the oracle: every vector in
tests/golden/ was produced by the reference itself (tests/golden/make_golden.py)."""

import numpy as np
import pytest

from conftest import assert_normwise, assert_stats, golden_stats, load_json, load_npz
from oracle.cvmatrix_oracle import (
    OracleCVMatrix,
    OraclePartitioner,
    benchmark_inputs,
    complement_indices,
    naive_training_matrices,
)

TOL = 1e-11  # oracle vs reference fast path, norm-wise


def check_case(z, key, X, Y, w, folds, flags, ddof, fold_subset=None, naive_atol=1e-8):
    m = OracleCVMatrix(*flags, ddof=ddof)
    m.fit(X, Y, w)
    p = OraclePartitioner(folds)
    labels = list(p.folds_dict) if fold_subset is None else fold_subset
    for fi, f in enumerate(labels):
        v = p.get_validation_indices(f)
        k = f"{key}/fold{fi}"
        assert np.array_equal(v, z[f"{k}/val"])
        if Y is not None:
            (xtx, xty), st = m.training_XTX_XTY(v)
            assert_normwise(xtx, z[f"{k}/fast/joint/XTX"], TOL, k + " XTX")
            assert_normwise(xty, z[f"{k}/fast/joint/XTY"], TOL, k + " XTY")
            assert_stats(st, golden_stats(z, f"{k}/fast/joint"), 1e-11, k)
            # the reference's own equivalence bar vs its naive implementation
            np.testing.assert_allclose(xtx, z[f"{k}/naive/joint/XTX"], atol=naive_atol)
            np.testing.assert_allclose(xty, z[f"{k}/naive/joint/XTY"], atol=naive_atol)
            xty2, st2 = m.training_XTY(v)
            assert np.array_equal(xty2, xty)
            assert_stats(st2, golden_stats(z, f"{k}/fast/xty"), 1e-11, k + " xty")
            # the oracle's own naive restatement agrees with the reference's naive
            t = complement_indices(p, f)
            (nx, ny), nst = naive_training_matrices(X, Y, w, t, *flags, ddof)
            np.testing.assert_allclose(nx, z[f"{k}/naive/joint/XTX"], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(ny, z[f"{k}/naive/joint/XTY"], rtol=1e-9, atol=1e-9)
        xtx1, st1 = m.training_XTX(v)
        ref = z[f"{k}/fast/joint/XTX"] if Y is not None else z[f"{k}/fast/xtx/XTX"]
        assert_normwise(xtx1, ref, TOL, k + " xtx-only")
        assert_stats(st1, golden_stats(z, f"{k}/fast/xtx"), 1e-11, k + " xtx")
        assert_stats(m.training_statistics(v), golden_stats(z, f"{k}/fast/stat"), 1e-11,
                     k + " stat")


def test_g1_inline_fixtures():
    z = load_npz("g1_inline.npz")
    X, Y, folds = z["X"], z["Y"], z["folds"]
    for i, w in enumerate(z["weights"]):
        check_case(z, f"w{i}", X, Y, w, folds, (True,) * 4, 1)
    check_case(z, "swapped", Y, X, None, folds, (True,) * 4, 1)
    # literal pin quoted in SURVEY.md 8c (weights [17,19,23,29,31], fold 0)
    m = OracleCVMatrix()
    m.fit(X, Y, z["weights"][0])
    (xtx, xty), (muX, sdX, muY, sdY) = m.training_XTX_XTY(np.array([0, 1]))
    np.testing.assert_allclose(xtx, [[55.33333333333333]], rtol=1e-13)
    np.testing.assert_allclose(xty, [[-55.33333333333342]], rtol=1e-13)
    np.testing.assert_allclose(muX, [[4.096385542168675]], rtol=1e-14)
    np.testing.assert_allclose(sdX, [[0.9807998548883993]], rtol=1e-13)
    np.testing.assert_allclose(muY, [[1.9036144578313252]], rtol=1e-14)


def test_g2_readme_quickstart():
    z = load_npz("g2_readme.npz")
    check_case(z, "c1", z["X"], z["Y"], z["w"], z["folds"], (True,) * 4, 1)


def test_g3_flag_sweep():
    z = load_npz("g3_sweep.npz")
    folds = z["folds"]
    for name in z["cases"]:
        name = str(name)
        if name.startswith("loo_"):
            fl = tuple(c == "1" for c in name[5:9])
            weighted = name.endswith("w1")
            check_case(z, name, z["Xw"] if weighted else z["X"], z["Y"],
                       z["w"] if weighted else None, np.arange(60), fl, 1,
                       fold_subset=list(range(20)))
        else:
            fl = tuple(c == "1" for c in name[1:5])
            weighted, ddof, hasY = name[7] == "1", int(name[10]), name[13] == "1"
            check_case(z, name, z["Xw"] if weighted else z["X"],
                       z["Y"] if hasY else None, z["w"] if weighted else None, folds, fl,
                       ddof)


def test_g4_example_zero_weight_and_str_label():
    z = load_npz("g4_example.npz")
    folds = load_json("g4_example_folds.json")
    m = OracleCVMatrix()
    m.fit(z["X"], z["Y"], z["w"])
    p = OraclePartitioner(folds)
    assert list(p.folds_dict) == [0, "one", 2]
    for i, k in enumerate(p.folds_dict):
        v = p.get_validation_indices(k)
        assert np.array_equal(v, z[f"fold{i}/val"])
        with np.errstate(all="ignore"):
            (xtx, xty), st = m.training_XTX_XTY(v)
        np.testing.assert_allclose(xtx, z[f"fold{i}/fast/joint/XTX"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(xty, z[f"fold{i}/fast/joint/XTY"], rtol=1e-9, atol=1e-9)
        assert_stats(st, golden_stats(z, f"fold{i}/fast/joint"), 1e-11, f"g4 fold{i}")


def _error_builders(CV, Part):
    X = np.arange(1.0, 13.0).reshape(6, 2)
    Y = np.arange(6.0)[::-1].copy()

    def m(w, ddof=1, flags=(True,) * 4, withY=True):
        c = CV(*flags, ddof=ddof)
        c.fit(X, Y if withY else None, w)
        return c

    w2 = np.array([1.0, 2.0, 0, 0, 0, 0])
    return {
        "negative_weight": lambda: CV().fit(X, Y, np.array([1, 1, -1, 1, 1, 1.0])),
        "ddof_joint": lambda: m(w2, 2).training_XTX_XTY(np.array([4, 5])),
        "ddof_xtx": lambda: m(w2, 2).training_XTX(np.array([4, 5])),
        "ddof_xty": lambda: m(w2, 2).training_XTY(np.array([4, 5])),
        "ddof_stat": lambda: m(w2, 2).training_statistics(np.array([4, 5])),
        "ddof_xtx_centerY_only_ok":
            lambda: m(w2, 2, (False, True, False, False)).training_XTX(np.array([4, 5])),
        "zero_joint": lambda: m(w2, 0).training_XTX_XTY(np.array([0, 1])),
        "zero_xtx": lambda: m(w2, 0).training_XTX(np.array([0, 1])),
        "zero_stat": lambda: m(w2, 0).training_statistics(np.array([0, 1])),
        "zero_noflags_ok":
            lambda: m(w2, 0, (False,) * 4).training_XTX_XTY(np.array([0, 1])),
        "zero_before_ddof": lambda: m(w2, 5).training_XTX_XTY(np.array([0, 1])),
        "noY_xty": lambda: m(None, 1, withY=False).training_XTY(np.array([0])),
        "noY_joint": lambda: m(None, 1, withY=False).training_XTX_XTY(np.array([0])),
        "neither": lambda: m(None)._training_matrices(False, False, np.array([0])),
        "unweighted_ddof": lambda: m(None, 4).training_XTX(np.array([0, 1])),
        "unweighted_all_val_scale": lambda: m(None, 0).training_XTX(np.arange(6)),
        "fold_missing": lambda: Part([0, 1, 1]).get_validation_indices(7),
        "fold_missing_str": lambda: Part([0, "a"]).get_validation_indices("b"),
    }


def test_g5_error_messages():
    gold = load_json("g5_errors.json")
    builders = _error_builders(OracleCVMatrix, OraclePartitioner)
    for name, fn in builders.items():
        exp = gold[name]
        if exp is None:
            with np.errstate(all="ignore"):
                fn()
            continue
        with pytest.raises(ValueError) as ei, np.errstate(all="ignore"):
            fn()
        assert type(ei.value).__name__ == exp[0] and str(ei.value) == exp[1], name


def test_g7_none_pattern():
    gold = load_json("g7_none.json")
    X = np.arange(1.0, 25.0).reshape(8, 3) ** 1.1
    Y = np.arange(16.0).reshape(8, 2) ** 0.9
    v = np.array([1, 5])
    pat = lambda st: "".join("x" if s is not None else "-" for s in st)  # noqa: E731
    for key, exp in gold.items():
        m = OracleCVMatrix(*[c == "1" for c in key])
        m.fit(X, Y)
        assert pat(m.training_XTX(v)[1]) == exp["xtx"]
        assert pat(m.training_XTY(v)[1]) == exp["xty"]
        assert pat(m.training_XTX_XTY(v)[1]) == exp["joint"]
        assert pat(m.training_statistics(v)) == exp["stat"]


def check_digest(z, name, f, xtx, xty, st, tol):
    k = f"{name}/fold{f}"
    sx, sy = z[f"{name}/samp_x"], z[f"{name}/samp_y"]
    x64, y64 = np.asarray(xtx, np.float64), np.asarray(xty, np.float64)
    mx, my = float(z[f"{k}/XTX_max"]), float(z[f"{k}/XTY_max"])
    assert abs(np.linalg.norm(x64) - z[f"{k}/XTX_fro"]) <= tol * z[f"{k}/XTX_fro"]
    assert abs(np.linalg.norm(y64) - z[f"{k}/XTY_fro"]) <= tol * z[f"{k}/XTY_fro"]
    assert np.abs(x64[sx[:, 0], sx[:, 1]] - z[f"{k}/XTX_samp"]).max() <= tol * mx
    assert np.abs(y64[sy[:, 0], sy[:, 1]] - z[f"{k}/XTY_samp"]).max() <= tol * my
    K = x64.shape[0]
    assert np.abs(x64.sum(axis=1) - z[f"{k}/XTX_rowsum"]).max() <= tol * mx * K
    assert np.abs(y64.sum(axis=0) - z[f"{k}/XTY_colsum"]).max() <= tol * my * K
    assert abs(np.trace(x64) - z[f"{k}/XTX_trace"]) <= tol * mx * K
    assert_stats(st, golden_stats(z, k), max(tol, 1e-10), k)


@pytest.mark.parametrize("name", ["c3", "c4s"])
def test_g6_digest_full_shapes(name):
    """BASELINE.json shapes: the oracle reproduces the reference's per-fold digests."""
    z = load_npz("g6_digest.npz")
    meta = load_json("g6_digest_meta.json")[name]
    X, Y, w, folds = benchmark_inputs(meta["N"], meta["K"], meta["M"], meta["P"])
    m = OracleCVMatrix(*meta["flags"], ddof=1, copy=False)
    m.fit(X, Y, w if meta["weighted"] else None)
    p = OraclePartitioner(folds)
    for f in z[f"{name}/fold_labels"][:3]:
        (xtx, xty), st = m.training_XTX_XTY(p.get_validation_indices(int(f)))
        check_digest(z, name, int(f), xtx, xty, st, 1e-10)
