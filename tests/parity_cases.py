"""Shared parity checks: run a CVMatrix-like model (the oracle on CPU, the HIP product on
the GPU) through the golden cases of tests/golden/ and compare with the reference's
outputs.  The structure follows the reference's own equivalence tests
(tests/test_cvmatrix.py:420-537: every fold x every method, fast vs naive)."""

import numpy as np
import pytest

from conftest import assert_normwise, assert_stats, golden_stats, load_json, load_npz, to_np


def check_case(z, key, make, part, X, Y, w, folds, flags, ddof, tol, fold_subset=None,
               naive_atol=1e-8, stat_rtol=None):
    """make(flags, ddof) -> model; part(folds) -> partitioner."""
    stat_rtol = stat_rtol or tol
    m = make(flags, ddof)
    m.fit(X, Y, w)
    p = part(folds)
    labels = list(p.folds_dict) if fold_subset is None else fold_subset
    for fi, f in enumerate(labels):
        v = p.get_validation_indices(f)
        k = f"{key}/fold{fi}"
        assert np.array_equal(v, z[f"{k}/val"])
        if Y is not None:
            (xtx, xty), st = m.training_XTX_XTY(v)
            assert_normwise(xtx, z[f"{k}/fast/joint/XTX"], tol, k + " XTX")
            assert_normwise(xty, z[f"{k}/fast/joint/XTY"], tol, k + " XTY")
            assert_stats(st, golden_stats(z, f"{k}/fast/joint"), stat_rtol, k)
            # the reference's own equivalence bar vs its naive implementation
            np.testing.assert_allclose(to_np(xtx), z[f"{k}/naive/joint/XTX"], atol=naive_atol)
            np.testing.assert_allclose(to_np(xty), z[f"{k}/naive/joint/XTY"], atol=naive_atol)
            xty2, st2 = m.training_XTY(v)
            assert_normwise(xty2, z[f"{k}/fast/joint/XTY"], tol, k + " xty-only")
            assert_stats(st2, golden_stats(z, f"{k}/fast/xty"), stat_rtol, k + " xty")
        xtx1, st1 = m.training_XTX(v)
        ref = z[f"{k}/fast/joint/XTX"] if Y is not None else z[f"{k}/fast/xtx/XTX"]
        assert_normwise(xtx1, ref, tol, k + " xtx-only")
        assert_stats(st1, golden_stats(z, f"{k}/fast/xtx"), stat_rtol, k + " xtx")
        assert_stats(m.training_statistics(v), golden_stats(z, f"{k}/fast/stat"), stat_rtol,
                     k + " stat")


def run_g1(make, part, tol):
    z = load_npz("g1_inline.npz")
    X, Y, folds = z["X"], z["Y"], z["folds"]
    for i, w in enumerate(z["weights"]):
        check_case(z, f"w{i}", make, part, X, Y, w, folds, (True,) * 4, 1, tol)
    check_case(z, "swapped", make, part, Y, X, None, folds, (True,) * 4, 1, tol)
    # literal pin quoted in SURVEY.md 8c (weights [17,19,23,29,31], fold 0)
    m = make((True,) * 4, 1)
    m.fit(X, Y, z["weights"][0])
    (xtx, xty), (muX, sdX, muY, sdY) = m.training_XTX_XTY(np.array([0, 1]))
    np.testing.assert_allclose(to_np(xtx), [[55.33333333333333]], rtol=1e-12)
    np.testing.assert_allclose(to_np(xty), [[-55.33333333333342]], rtol=1e-12)
    np.testing.assert_allclose(to_np(muX), [[4.096385542168675]], rtol=1e-13)
    np.testing.assert_allclose(to_np(sdX), [[0.9807998548883993]], rtol=1e-12)
    np.testing.assert_allclose(to_np(muY), [[1.9036144578313252]], rtol=1e-13)


def run_g2(make, part, tol):
    z = load_npz("g2_readme.npz")
    check_case(z, "c1", make, part, z["X"], z["Y"], z["w"], z["folds"], (True,) * 4, 1, tol)


def g3_cases():
    z = load_npz("g3_sweep.npz")
    return [str(c) for c in z["cases"]]


def run_g3_case(name, make, part, tol):
    z = load_npz("g3_sweep.npz")
    if name.startswith("loo_"):
        fl = tuple(c == "1" for c in name[5:9])
        weighted = name.endswith("w1")
        check_case(z, name, make, part, z["Xw"] if weighted else z["X"], z["Y"],
                   z["w"] if weighted else None, np.arange(60), fl, 1, tol,
                   fold_subset=list(range(20)))
    else:
        fl = tuple(c == "1" for c in name[1:5])
        weighted, ddof, hasY = name[7] == "1", int(name[10]), name[13] == "1"
        check_case(z, name, make, part, z["Xw"] if weighted else z["X"],
                   z["Y"] if hasY else None, z["w"] if weighted else None, z["folds"], fl,
                   ddof, tol)


def g3loo_cases():
    z = load_npz("g3_loo.npz")
    return [str(c) for c in z["cases"]]


def _stacked_stats(z, key, f):
    mask = z[f"{key}/mask"]
    return tuple(z[f"{key}/{n}"][f].reshape(1, -1) if m else None for n, m in zip(("muX", "sdX", "muY", "sdY"), mask))


def run_g3loo_case(name, make, tol, batched=False):
    """One configuration of the reference's leave-one-out sweep (tests/test_cvmatrix.py:1357-1396;
    tests/golden/g3_loo.npz): every method for the first 20 one-sample folds, against the reference's
    fast outputs (norm-wise ``tol``) and its naive outputs (the reference's own atol 1e-8).
    ``batched``: ask for the 20 folds in one ``*_batched`` call (product only)."""
    z = load_npz("g3_loo.npz")
    g = load_npz("g3_sweep.npz")
    fl = tuple(c == "1" for c in name[5:9])
    weighted, ddof, hasY = name[11] == "1", int(name[14]), name[17] == "1"
    X = g["Xw"] if weighted else g["X"]
    Y = g["Y"] if hasY else None
    w = g["w"] if weighted else None
    m = make(fl, ddof)
    m.fit(X, Y, w)
    nf = z[f"{name}/fast_XTX"].shape[0]
    if batched:
        folds = [np.array([f]) for f in range(nf)]
        if hasY:
            (bx, by), bst = m.training_XTX_XTY_batched(folds)
            _, bst_y = m.training_XTY_batched(folds)
        bx1, bst_x = m.training_XTX_batched(folds)
        bstat = m.training_statistics_batched(folds)
        pick = lambda st, f: tuple(None if s is None else s[f] for s in st)  # noqa: E731
    for f in range(nf):
        v = np.array([f])
        k = f"{name} fold {f}"
        if hasY:
            (xtx, xty), st = ((bx[f], by[f]), pick(bst, f)) if batched else m.training_XTX_XTY(v)
            assert_normwise(xtx, z[f"{name}/fast_XTX"][f], tol, k + " XTX")
            assert_normwise(xty, z[f"{name}/fast_XTY"][f], tol, k + " XTY")
            assert_stats(st, _stacked_stats(z, f"{name}/joint", f), tol, k)
            np.testing.assert_allclose(to_np(xtx), z[f"{name}/naive_XTX"][f], atol=1e-8)
            np.testing.assert_allclose(to_np(xty), z[f"{name}/naive_XTY"][f], atol=1e-8)
            st_y = pick(bst_y, f) if batched else m.training_XTY(v)[1]
            assert_stats(st_y, _stacked_stats(z, f"{name}/xty", f), tol, k + " xty")
        xtx1, st1 = (bx1[f], pick(bst_x, f)) if batched else m.training_XTX(v)
        assert_normwise(xtx1, z[f"{name}/fast_XTX"][f], tol, k + " xtx-only")
        assert_stats(st1, _stacked_stats(z, f"{name}/xtx", f), tol, k + " xtx")
        if not hasY:
            np.testing.assert_allclose(to_np(xtx1), z[f"{name}/naive_XTX"][f], atol=1e-8)
        st4 = pick(bstat, f) if batched else m.training_statistics(v)
        assert_stats(st4, _stacked_stats(z, f"{name}/stat", f), tol, k + " stat")


def run_g4(make, part, tol):
    z = load_npz("g4_example.npz")
    folds = load_json("g4_example_folds.json")
    m = make((True,) * 4, 1)
    m.fit(z["X"], z["Y"], z["w"])
    p = part(folds)
    assert list(p.folds_dict) == [0, "one", 2]
    for i, k in enumerate(p.folds_dict):
        v = p.get_validation_indices(k)
        assert np.array_equal(v, z[f"fold{i}/val"])
        with np.errstate(all="ignore"):
            (xtx, xty), st = m.training_XTX_XTY(v)
        np.testing.assert_allclose(to_np(xtx), z[f"fold{i}/fast/joint/XTX"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(to_np(xty), z[f"fold{i}/fast/joint/XTY"], rtol=1e-9, atol=1e-9)
        assert_stats(st, golden_stats(z, f"fold{i}/fast/joint"), 1e-10, f"g4 fold{i}")


def error_builders(CV, Part):
    """The call sequences of tests/golden/make_golden.py::g5_errors."""
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


def run_g5(CV, Part):
    gold = load_json("g5_errors.json")
    for name, fn in error_builders(CV, Part).items():
        exp = gold[name]
        if exp is None:
            with np.errstate(all="ignore"):
                fn()
            continue
        with pytest.raises(ValueError) as ei, np.errstate(all="ignore"):
            fn()
        assert type(ei.value).__name__ == exp[0] and str(ei.value) == exp[1], name


def run_g7(CV):
    gold = load_json("g7_none.json")
    X = np.arange(1.0, 25.0).reshape(8, 3) ** 1.1
    Y = np.arange(16.0).reshape(8, 2) ** 0.9
    v = np.array([1, 5])
    pat = lambda st: "".join("x" if s is not None else "-" for s in st)  # noqa: E731
    for key, exp in gold.items():
        m = CV(*[c == "1" for c in key])
        m.fit(X, Y)
        assert pat(m.training_XTX(v)[1]) == exp["xtx"], key
        assert pat(m.training_XTY(v)[1]) == exp["xty"], key
        assert pat(m.training_XTX_XTY(v)[1]) == exp["joint"], key
        assert pat(m.training_statistics(v)) == exp["stat"], key


def check_digest(z, name, f, xtx, xty, st, tol, stat_rtol=None):
    """Compare one fold's result with the reference's stored digest (g6_digest.npz)."""
    k = f"{name}/fold{f}"
    sx, sy = z[f"{name}/samp_x"], z[f"{name}/samp_y"]
    x64, y64 = to_np(xtx).astype(np.float64), to_np(xty).astype(np.float64)
    mx, my = float(z[f"{k}/XTX_max"]), float(z[f"{k}/XTY_max"])
    assert abs(np.linalg.norm(x64) - z[f"{k}/XTX_fro"]) <= tol * z[f"{k}/XTX_fro"]
    assert abs(np.linalg.norm(y64) - z[f"{k}/XTY_fro"]) <= tol * z[f"{k}/XTY_fro"]
    assert np.abs(x64[sx[:, 0], sx[:, 1]] - z[f"{k}/XTX_samp"]).max() <= tol * mx
    assert np.abs(y64[sy[:, 0], sy[:, 1]] - z[f"{k}/XTY_samp"]).max() <= tol * my
    K = x64.shape[0]
    assert np.abs(x64.sum(axis=1) - z[f"{k}/XTX_rowsum"]).max() <= tol * mx * K
    assert np.abs(y64.sum(axis=0) - z[f"{k}/XTY_colsum"]).max() <= tol * my * K
    assert abs(np.trace(x64) - z[f"{k}/XTX_trace"]) <= tol * mx * K
    assert_stats(st, golden_stats(z, k), stat_rtol or max(tol, 1e-10), k)
    check_strips(name, f, x64, y64, tol)


_STRIPS = None


def check_strips(name, f, x64, y64, tol):
    """Whole rows of the reference's XTX (64 of them, spread over the matrix) and its whole XTY
    (tests/golden/g6_strips.npz, the folds it holds): the norm of the DIFFERENCE, max-norm and
    Frobenius, against ``tol`` times the reference's."""
    global _STRIPS
    if _STRIPS is None:
        _STRIPS = load_npz("g6_strips.npz")
    zs = _STRIPS
    key = f"{name}/fold{f}/XTX_rows"
    if key not in zs.files:
        return False
    rows, rx, ry = zs[f"{name}/rows"], zs[key], zs[f"{name}/fold{f}/XTY"]
    for got, ref, what in ((x64[rows], rx, "XTX rows"), (y64, ry, "XTY")):
        d = got - ref
        assert np.abs(d).max() <= tol * np.abs(ref).max(), f"{name} fold {f} {what}: max {np.abs(d).max():.3e}"
        assert np.linalg.norm(d) <= tol * np.linalg.norm(ref), f"{name} fold {f} {what}: Frobenius"
    return True
