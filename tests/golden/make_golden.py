"""
Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE
(sm00thix/cvmatrix v3.2.1 mounted read-only at /root/reference).

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Outputs are data only: the inputs (or the seed that regenerates them) and the outputs of
the reference's ``CVMatrix(backend="numpy")`` ("fast") and of its test oracle
``tests/naive_cvmatrix.py::NaiveCVMatrix`` ("naive").  Groups:

  g1_inline.npz    the five inline fixtures of tests/test_cvmatrix.py
                   (1025-1028, 1089-1092, 1157-1160, 1217-1220, 1256-1259)
  g2_readme.npz    README.md:96-141 quick-start shape (N=100,K=50,M=10, 5 folds, w+0.1)
  g3_sweep.npz     16 flag combos x {weighted(10% zeros), unweighted} x ddof{0,1} x
                   {Y, None}, N=60,K=8,M=3, 3 uneven folds (+ LOOCV on a subset);
                   mirrors tests/test_cvmatrix.py:539-575, 1357-1396
  g4_example.npz   examples/training_matrices.py:19-21,30 (zero weight, str fold label)
  g5_errors.json   error cases -> (exception type, message)
  g6_digest.npz    per-fold digests at the BASELINE.json shapes C2/C3 and scaled C4/C5,
                   inputs from default_rng(42) exactly as benchmarks/benchmark.py:223-233
  g7_none.json     which statistics come back None
  g3_loo.npz       the reference's whole leave-one-out sweep (tests/test_cvmatrix.py:1357-1396):
                   16 flag combos x {weighted, unweighted} x ddof{0,1} x {Y, None}, first 20 folds
  g6_strips.npz    for a few folds of the g6 configurations: 64 whole rows of XTX (16 for the
                   K=4096 case) and the whole XTY, so that the comparison is the norm of the
                   DIFFERENCE over a strip, not a difference of norms
"""

import itertools
import json
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from cvmatrix import CVMatrix, Partitioner  # noqa: E402
from tests.naive_cvmatrix import NaiveCVMatrix  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
STAT_NAMES = ("muX", "sdX", "muY", "sdY")
FLAGS16 = list(itertools.product([False, True], repeat=4))


def train_idx(p, fold):
    parts = [p.get_validation_indices(f) for f in p.folds_dict if f != fold]
    return np.concatenate(parts) if parts else np.zeros((0,), dtype=int)


def put_stats(out, key, stats):
    mask = np.zeros(4, dtype=bool)
    for i, (n, s) in enumerate(zip(STAT_NAMES, stats)):
        if s is not None:
            mask[i] = True
            out[f"{key}/{n}"] = np.asarray(s)
    out[f"{key}/mask"] = mask


def put_joint(out, key, res):
    (xtx, xty), stats = res
    out[f"{key}/XTX"] = xtx
    out[f"{key}/XTY"] = xty
    put_stats(out, key, stats)


def run_case(out, key, X, Y, w, folds, flags, ddof, dtype=np.float64, fold_subset=None):
    """Store fast+naive outputs of every method for every fold of one configuration."""
    cX, cY, sX, sY = flags
    fast = CVMatrix(cX, cY, sX, sY, ddof, dtype, True, backend="numpy")
    naive = NaiveCVMatrix(cX, cY, sX, sY, ddof, dtype, True)
    fast.fit(X, Y, w)
    naive.fit(X, Y, w)
    p = Partitioner(folds)
    labels = list(p.folds_dict) if fold_subset is None else fold_subset
    for fi, f in enumerate(labels):
        v = p.get_validation_indices(f)
        t = train_idx(p, f)
        k = f"{key}/fold{fi}"
        out[f"{k}/val"] = v
        if Y is not None:
            put_joint(out, f"{k}/fast/joint", fast.training_XTX_XTY(v))
            put_joint(out, f"{k}/naive/joint", naive.training_XTX_XTY(t))
            m, st = fast.training_XTY(v)
            assert np.array_equal(m, out[f"{k}/fast/joint/XTY"])
            put_stats(out, f"{k}/fast/xty", st)
        m, st = fast.training_XTX(v)
        if Y is None:
            out[f"{k}/fast/xtx/XTX"] = m
        else:  # same computation as the joint call (cvmatrix.py:843-853 vs 870-880)
            assert np.array_equal(m, out[f"{k}/fast/joint/XTX"])
        put_stats(out, f"{k}/fast/xtx", st)
        m, st = naive.training_XTX(t)
        if Y is None:
            out[f"{k}/naive/xtx/XTX"] = m
        put_stats(out, f"{k}/naive/xtx", st)
        put_stats(out, f"{k}/fast/stat", fast.training_statistics(v))


def g1_inline():
    out = {}
    X = np.array([1, 2, 3, 4, 5])
    Y = np.array([5, 4, 3, 2, 1])
    folds = np.array([0, 0, 1, 1, 2])
    out["X"], out["Y"], out["folds"] = X, Y, folds
    ws = [[17, 19, 23, 29, 31], [2, 4, 6, 8, 10], [3, 6, 9, 12, 15], [2, 5, 7, 11, 13],
          [37, 41, 43, 47, 53]]
    out["weights"] = np.array(ws)
    for i, w in enumerate(ws):
        run_case(out, f"w{i}", X, Y, np.array(w), folds, (True,) * 4, 1)
    # test_switch_matrices (1020-1043): re-fit with X and Y swapped, unweighted
    run_case(out, "swapped", Y, X, None, folds, (True,) * 4, 1)
    np.savez_compressed(os.path.join(HERE, "g1_inline.npz"), **out)


def g2_readme():
    out = {}
    rng = np.random.default_rng(42)
    N, K, M = 100, 50, 10
    X = rng.random((N, K))
    Y = rng.random((N, M))
    w = rng.random((N,)) + 0.1
    folds = np.arange(N) % 5
    out.update(X=X, Y=Y, w=w, folds=folds)
    run_case(out, "c1", X, Y, w, folds, (True,) * 4, 1)
    np.savez_compressed(os.path.join(HERE, "g2_readme.npz"), **out)


def g3_sweep():
    out = {}
    rng = np.random.default_rng(20240601)
    N, K, M = 60, 8, 3
    X = rng.random((N, K)) * 3.0 + rng.random((1, K))
    X[:, 5] = 1.0  # constant column; exact when unweighted (tests 1045-1081)
    Y = rng.standard_normal((N, M)) + 2.0
    w = rng.random((N,)) + 0.05
    w[rng.choice(N, size=N // 10, replace=False)] = 0.0
    folds = rng.permutation(np.repeat([0, 1, 2], [30, 20, 10]))
    # weighted runs use a copy of X without the constant column pinned
    Xw = X.copy()
    Xw[:, 5] = rng.random(N)
    out.update(X=X, Xw=Xw, Y=Y, w=w, folds=folds)
    loocv = np.arange(N)
    loocv_subset = list(range(20))
    cases = []
    for flags in FLAGS16:
        for weighted in (False, True):
            for ddof in (0, 1):
                for hasY in (True, False):
                    name = "f{}{}{}{}_w{}_d{}_y{}".format(
                        *[int(b) for b in flags], int(weighted), ddof, int(hasY))
                    cases.append(name)
                    run_case(out, name, Xw if weighted else X, Y if hasY else None,
                             w if weighted else None, folds, flags, ddof)
    for flags in [(False,) * 4, (True,) * 4, (True, False, False, True), (False, True, True, False)]:
        for weighted in (False, True):
            name = "loo_f{}{}{}{}_w{}".format(*[int(b) for b in flags], int(weighted))
            cases.append(name)
            run_case(out, name, Xw if weighted else X, Y, w if weighted else None,
                     loocv, flags, 1, fold_subset=loocv_subset)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "g3_sweep.npz"), **out)


def stack_stats(stats_list):
    """[(muX, sdX, muY, sdY)] of several folds -> (mask[4], one 2-D array per present statistic)."""
    mask = np.array([s is not None for s in stats_list[0]])
    arrs = [np.stack([np.asarray(st[i]).reshape(-1) for st in stats_list]) if mask[i] else None
            for i in range(4)]
    return mask, arrs


def put_stacked_stats(out, key, stats_list):
    mask, arrs = stack_stats(stats_list)
    out[f"{key}/mask"] = mask
    for n, a in zip(STAT_NAMES, arrs):
        if a is not None:
            out[f"{key}/{n}"] = a


def g3_loo():
    """The reference's leave-one-out sweep (tests/test_cvmatrix.py:1357-1396), on the g3 inputs:
    every flag combination x weights x ddof x Y/None, folds = arange(N), the first 20 folds.
    Stored stacked over the folds (one array per case and quantity)."""
    z = np.load(os.path.join(HERE, "g3_sweep.npz"))
    X, Xw, Y, w = z["X"], z["Xw"], z["Y"], z["w"]
    N = X.shape[0]
    out, cases = {}, []
    nf = 20
    for flags in FLAGS16:
        for weighted in (False, True):
            for ddof in (0, 1):
                for hasY in (True, False):
                    name = "loo_f{}{}{}{}_w{}_d{}_y{}".format(
                        *[int(b) for b in flags], int(weighted), ddof, int(hasY))
                    cases.append(name)
                    Xc, Yc, wc = (Xw if weighted else X), (Y if hasY else None), (w if weighted else None)
                    cX, cY, sX, sY = flags
                    fast = CVMatrix(cX, cY, sX, sY, ddof, np.float64, True, backend="numpy")
                    naive = NaiveCVMatrix(cX, cY, sX, sY, ddof, np.float64, True)
                    fast.fit(Xc, Yc, wc)
                    naive.fit(Xc, Yc, wc)
                    fx, fy, nx, ny, st_joint, st_xtx, st_xty, st_stat = [], [], [], [], [], [], [], []
                    for f in range(nf):
                        v = np.array([f])
                        t = np.delete(np.arange(N), f)
                        if hasY:
                            (a, b), s1 = fast.training_XTX_XTY(v)
                            (c, d), _ = naive.training_XTX_XTY(t)
                            fx.append(a); fy.append(b); nx.append(c); ny.append(d); st_joint.append(s1)
                            st_xty.append(fast.training_XTY(v)[1])
                        a, s2 = fast.training_XTX(v)
                        st_xtx.append(s2)
                        if not hasY:
                            fx.append(a); nx.append(naive.training_XTX(t)[0])
                        st_stat.append(fast.training_statistics(v))
                    out[f"{name}/fast_XTX"], out[f"{name}/naive_XTX"] = np.stack(fx), np.stack(nx)
                    if hasY:
                        out[f"{name}/fast_XTY"], out[f"{name}/naive_XTY"] = np.stack(fy), np.stack(ny)
                        put_stacked_stats(out, f"{name}/joint", st_joint)
                        put_stacked_stats(out, f"{name}/xty", st_xty)
                    put_stacked_stats(out, f"{name}/xtx", st_xtx)
                    put_stacked_stats(out, f"{name}/stat", st_stat)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "g3_loo.npz"), **out)


def g4_example():
    out = {}
    X = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9], [10, 11, 12]])
    Y = np.array([[1, 2], [3, 4], [5, 6], [7, 8]])
    w = np.array([4.2, 13.37, 3.14, 0])
    folds = [0, "one", 2, 2]
    out.update(X=X, Y=Y, w=w)
    with open(os.path.join(HERE, "g4_example_folds.json"), "w") as f:
        json.dump(folds, f)
    fast = CVMatrix(True, True, True, True)
    fast.fit(X, Y, w)
    p = Partitioner(folds)
    keys = list(p.folds_dict)
    assert keys == [0, "one", 2]
    for i, k in enumerate(keys):
        v = p.get_validation_indices(k)
        out[f"fold{i}/val"] = v
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            put_joint(out, f"fold{i}/fast/joint", fast.training_XTX_XTY(v))
    np.savez_compressed(os.path.join(HERE, "g4_example.npz"), **out)


def g5_errors():
    """(name, how to build it) -> exception type + message, captured from the reference."""
    X = np.arange(1.0, 13.0).reshape(6, 2)
    Y = np.arange(6.0)[::-1].copy()
    res = {}

    def cap(name, fn):
        try:
            fn()
            res[name] = None
        except Exception as e:  # noqa: BLE001
            res[name] = [type(e).__name__, str(e)]

    def neg():
        CVMatrix().fit(X, Y, np.array([1, 1, -1, 1, 1, 1.0]))

    def m(w, ddof=1, flags=(True,) * 4, withY=True):
        c = CVMatrix(*flags, ddof=ddof)
        c.fit(X, Y if withY else None, w)
        return c

    cap("negative_weight", neg)
    w2 = np.array([1.0, 2.0, 0, 0, 0, 0])
    cap("ddof_joint", lambda: m(w2, 2).training_XTX_XTY(np.array([4, 5])))
    cap("ddof_xtx", lambda: m(w2, 2).training_XTX(np.array([4, 5])))
    cap("ddof_xty", lambda: m(w2, 2).training_XTY(np.array([4, 5])))
    cap("ddof_stat", lambda: m(w2, 2).training_statistics(np.array([4, 5])))
    cap("ddof_xtx_centerY_only_ok",
        lambda: m(w2, 2, (False, True, False, False)).training_XTX(np.array([4, 5])))
    cap("zero_joint", lambda: m(w2, 0).training_XTX_XTY(np.array([0, 1])))
    cap("zero_xtx", lambda: m(w2, 0).training_XTX(np.array([0, 1])))
    cap("zero_stat", lambda: m(w2, 0).training_statistics(np.array([0, 1])))
    cap("zero_noflags_ok", lambda: m(w2, 0, (False,) * 4).training_XTX_XTY(np.array([0, 1])))
    cap("zero_before_ddof", lambda: m(w2, 5).training_XTX_XTY(np.array([0, 1])))
    cap("noY_xty", lambda: m(None, 1, withY=False).training_XTY(np.array([0])))
    cap("noY_joint", lambda: m(None, 1, withY=False).training_XTX_XTY(np.array([0])))
    cap("neither", lambda: m(None)._training_matrices(False, False, np.array([0])))
    cap("unweighted_ddof", lambda: m(None, 4).training_XTX(np.array([0, 1])))
    cap("unweighted_all_val_scale",
        lambda: m(None, 0).training_XTX(np.arange(6)))
    cap("fold_missing", lambda: Partitioner([0, 1, 1]).get_validation_indices(7))
    cap("fold_missing_str", lambda: Partitioner([0, "a"]).get_validation_indices("b"))
    cap("bad_backend", lambda: CVMatrix(backend="tpu"))
    with open(os.path.join(HERE, "g5_errors.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


def digest(out, key, xtx, xty, stats, samp_x, samp_y):
    out[f"{key}/XTX_fro"] = np.linalg.norm(xtx.astype(np.float64))
    out[f"{key}/XTX_max"] = np.abs(xtx).max()
    out[f"{key}/XTX_trace"] = np.trace(xtx.astype(np.float64))
    out[f"{key}/XTX_samp"] = xtx[samp_x[:, 0], samp_x[:, 1]]
    out[f"{key}/XTX_rowsum"] = xtx.astype(np.float64).sum(axis=1)
    out[f"{key}/XTY_fro"] = np.linalg.norm(xty.astype(np.float64))
    out[f"{key}/XTY_max"] = np.abs(xty).max()
    out[f"{key}/XTY_samp"] = xty[samp_y[:, 0], samp_y[:, 1]]
    out[f"{key}/XTY_colsum"] = xty.astype(np.float64).sum(axis=0)
    put_stats(out, key, stats)


def g6_digest():
    """name, N, K, M, P, weighted, flags, dtype, dtype of the reference run.
    For fp32 (C5) the pinned outputs are the reference run in FLOAT64 on the float32
    inputs (the parity rule of SURVEY 8d compares fp32 results to the fp64 result) plus
    the reference's own fp32 error norm as the yardstick."""
    out = {}
    cfgs = [
        ("c2", 100000, 512, 16, 10, False, (False,) * 4, np.float64),
        ("c3", 100000, 512, 16, 10, True, (True,) * 4, np.float64),
        ("c4s", 20000, 1024, 32, 64, True, (True,) * 4, np.float64),
        ("c5s", 8000, 4096, 1, 20, True, (True,) * 4, np.float32),
    ]
    srng = np.random.default_rng(123)
    meta = {}
    for name, N, K, M, P, weighted, flags, dt in cfgs:
        rng = np.random.default_rng(42)
        X = rng.random((N, K), dtype=dt)
        Y = rng.random((N, M), dtype=dt)
        w = rng.random((N,), dtype=dt)
        folds = np.arange(N) % P
        samp_x = srng.integers(0, K, size=(64, 2))
        samp_y = np.stack([srng.integers(0, K, size=64), srng.integers(0, M, size=64)], 1)
        out[f"{name}/samp_x"], out[f"{name}/samp_y"] = samp_x, samp_y
        meta[name] = dict(N=N, K=K, M=M, P=P, weighted=weighted, flags=list(flags),
                          dtype=np.dtype(dt).name)
        ref = CVMatrix(*flags, ddof=1, dtype=np.float64, copy=False)
        ref.fit(X.astype(np.float64), Y.astype(np.float64),
                w.astype(np.float64) if weighted else None)
        ref32 = None
        if dt is np.float32:
            ref32 = CVMatrix(*flags, ddof=1, dtype=np.float32, copy=False)
            ref32.fit(X, Y, w if weighted else None)
        p = Partitioner(folds)
        fl = list(p.folds_dict)
        fl = fl if len(fl) <= 10 else fl[:4] + fl[-2:]
        out[f"{name}/fold_labels"] = np.array(fl)
        for f in fl:
            v = p.get_validation_indices(f)
            (xtx, xty), st = ref.training_XTX_XTY(v)
            digest(out, f"{name}/fold{f}", xtx, xty, st, samp_x, samp_y)
            if ref32 is not None:
                (x32, y32), _ = ref32.training_XTX_XTY(v)
                out[f"{name}/fold{f}/ref32_XTX_relfro"] = (
                    np.linalg.norm(x32.astype(np.float64) - xtx) / np.linalg.norm(xtx))
                out[f"{name}/fold{f}/ref32_XTY_relfro"] = (
                    np.linalg.norm(y32.astype(np.float64) - xty) / np.linalg.norm(xty))
        print("g6", name, "done", flush=True)
    with open(os.path.join(HERE, "g6_digest_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "g6_digest.npz"), **out)


def g6_strips():
    """Whole rows of the reference's outputs at the g6 configurations (same inputs, same runs)."""
    out = {}
    cfgs = [
        ("c2", 100000, 512, 16, 10, False, (False,) * 4, np.float64, (0, 4, 9), 64),
        ("c3", 100000, 512, 16, 10, True, (True,) * 4, np.float64, (0, 4, 9), 64),
        ("c4s", 20000, 1024, 32, 64, True, (True,) * 4, np.float64, (0, 63), 64),
        ("c5s", 8000, 4096, 1, 20, True, (True,) * 4, np.float32, (0, 19), 16),
    ]
    for name, N, K, M, P, weighted, flags, dt, folds_kept, nrows in cfgs:
        rng = np.random.default_rng(42)
        X = rng.random((N, K), dtype=dt)
        Y = rng.random((N, M), dtype=dt)
        w = rng.random((N,), dtype=dt)
        folds = np.arange(N) % P
        rows = np.unique(np.linspace(0, K - 1, nrows).round().astype(np.int64))
        out[f"{name}/rows"] = rows
        ref = CVMatrix(*flags, ddof=1, dtype=np.float64, copy=False)
        ref.fit(X.astype(np.float64), Y.astype(np.float64), w.astype(np.float64) if weighted else None)
        p = Partitioner(folds)
        for f in folds_kept:
            (xtx, xty), _ = ref.training_XTX_XTY(p.get_validation_indices(f))
            out[f"{name}/fold{f}/XTX_rows"] = xtx[rows]
            out[f"{name}/fold{f}/XTY"] = xty
        print("g6 strips", name, "done", flush=True)
    np.savez_compressed(os.path.join(HERE, "g6_strips.npz"), **out)


def g7_none():
    X = np.arange(1.0, 25.0).reshape(8, 3) ** 1.1
    Y = np.arange(16.0).reshape(8, 2) ** 0.9
    v = np.array([1, 5])
    res = {}
    for flags in FLAGS16:
        c = CVMatrix(*flags)
        c.fit(X, Y)
        key = "".join(str(int(b)) for b in flags)
        pat = lambda st: "".join("x" if s is not None else "-" for s in st)  # noqa: E731
        res[key] = dict(
            xtx=pat(c.training_XTX(v)[1]),
            xty=pat(c.training_XTY(v)[1]),
            joint=pat(c.training_XTX_XTY(v)[1]),
            stat=pat(c.training_statistics(v)),
        )
    with open(os.path.join(HERE, "g7_none.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g7", "g6"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for g in which:
            dict(g1=g1_inline, g2=g2_readme, g3=g3_sweep, g4=g4_example, g5=g5_errors,
                 g6=g6_digest, g7=g7_none, g3loo=g3_loo, g6strips=g6_strips)[g]()
            print("wrote", g, flush=True)
