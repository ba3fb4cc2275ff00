"""
CPU oracle for the cvmatrix hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This module is a from-scratch NumPy restatement of the algorithm that the reference
(sm00thix/cvmatrix v3.2.1, ``cvmatrix/cvmatrix.py``, ``cvmatrix/partitioner.py`` and the
test-only ``tests/naive_cvmatrix.py``) executes on the per-fold training-matrix path.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import it; nothing under ``cvmatrix_amd/`` does.  It never touches a GPU.

Parity status: PINNED.  The restatement is checked (tests/test_oracle_golden.py) against
golden vectors in ``tests/golden/*.npz`` that were produced in the build container by
importing the reference itself (``tests/golden/make_golden.py`` is the generating
script; the reference Python cannot travel to the GPU box, the vectors do).

The arithmetic lives in a third-party dependency of the reference, NumPy
(``numpy>=2.0,<3`` in the reference's pyproject.toml:16-18; 2.2.6 + OpenBLAS 0.3.29 in
the build container).  The formulation below is *fused*: it never keeps the N x K
temporaries ``WX = X*w`` and ``sq_X = WX*X`` of the reference alive (cvmatrix.py:1205,
1235); the full-data Gram is accumulated over row blocks instead.  Results therefore
agree with the reference to rounding (about 1e-13 norm-wise), not bit for bit.

Symbols (reference line numbers are for cvmatrix/cvmatrix.py unless stated):

    G  = X^T W X   (K,K)     H  = X^T W Y   (K,M)
    sX = sum_i w_i x_i       sY = sum_i w_i y_i
    qX = sum_i w_i x_i^2     qY = sum_i w_i y_i^2
    sw = sum_i w_i           nz = #{w_i != 0}
"""

from __future__ import annotations

from collections.abc import Hashable
from typing import Iterable, Optional

import numpy as np

ROW_BLOCK = 8192  # rows per accumulation block in fit_globals


# --------------------------------------------------------------------------------------
# Partitioner  (reference: cvmatrix/partitioner.py:48-107)
# --------------------------------------------------------------------------------------
class OraclePartitioner:
    """Fold label -> int index array, keys in first-seen order, indices ascending.

    Restates cvmatrix/partitioner.py:89-107 (``_init_folds_dict``) and :61-87
    (``get_validation_indices`` incl. the ``ValueError("Fold ... not found.")``).
    """

    def __init__(self, folds: Iterable[Hashable]) -> None:
        buckets: dict = {}
        for pos, label in enumerate(folds):
            buckets.setdefault(label, []).append(pos)
        self.folds_dict = {k: np.asarray(v, dtype=int) for k, v in buckets.items()}

    def get_validation_indices(self, fold: Hashable) -> np.ndarray:
        if fold not in self.folds_dict:
            raise ValueError(f"Fold {fold} not found.")
        return self.folds_dict[fold]


# --------------------------------------------------------------------------------------
# fit stage  (reference: cvmatrix.py:1131-1243)
# --------------------------------------------------------------------------------------
def _as_2d(a, dtype, copy: bool) -> np.ndarray:
    """cvmatrix.py:1146-1151: cast, copy iff requested, 1-D -> (N,1)."""
    out = np.asarray(a, dtype=dtype)
    if copy:
        out = out.copy()
    if out.ndim == 1:
        out = out.reshape(-1, 1)
    return out


def fit_globals(X, Y, w, center_X, center_Y, scale_X, scale_Y):
    """Full-data Gram and column statistics, accumulated over row blocks.

    Restates ``_init_weighted_mats`` (1193-1207), ``_init_matrix_products`` (1209-1217)
    and ``_init_stats`` (1219-1243).  Every statistic is produced only under the flag
    condition the reference uses (1223, 1230, 1232, 1234, 1239); the others stay None.
    """
    N, K = X.shape
    dt = X.dtype
    anyflag = center_X or center_Y or scale_X or scale_Y
    want_sX = center_X or center_Y or scale_X
    want_sY = (center_X or center_Y or scale_Y) and Y is not None
    want_qX = scale_X
    want_qY = scale_Y and Y is not None
    G = np.zeros((K, K), dtype=dt)
    H = np.zeros((K, Y.shape[1]), dtype=dt) if Y is not None else None
    sX = np.zeros((1, K), dtype=dt) if want_sX else None
    sY = np.zeros((1, Y.shape[1]), dtype=dt) if want_sY else None
    qX = np.zeros((1, K), dtype=dt) if want_qX else None
    qY = np.zeros((1, Y.shape[1]), dtype=dt) if want_qY else None
    for r0 in range(0, N, ROW_BLOCK):
        Xb = X[r0 : r0 + ROW_BLOCK]
        Yb = Y[r0 : r0 + ROW_BLOCK] if Y is not None else None
        if w is None:
            WXb, WYb = Xb, Yb
        else:
            wb = w[r0 : r0 + ROW_BLOCK]
            WXb = Xb * wb
            WYb = Yb * wb if (want_sY or want_qY) else None
        G += WXb.T @ Xb
        if Y is not None:
            H += WXb.T @ Yb
        if want_sX:
            sX += WXb.sum(axis=0, keepdims=True)
        if want_sY:
            sY += WYb.sum(axis=0, keepdims=True)
        if want_qX:
            qX += (WXb * Xb).sum(axis=0, keepdims=True)
        if want_qY:
            qY += (WYb * Yb).sum(axis=0, keepdims=True)
    if anyflag:
        if w is None:
            sw, nz = N, N  # python ints, cvmatrix.py:1227-1229
        else:
            sw, nz = np.sum(w), np.count_nonzero(w)  # cvmatrix.py:1225-1226
    else:
        sw = nz = None
    return dict(G=G, H=H, sX=sX, sY=sY, qX=qX, qY=qY, sw=sw, nz=nz)


# --------------------------------------------------------------------------------------
# fold stage  (reference: cvmatrix.py:589-1129)
# --------------------------------------------------------------------------------------
MSG_NEG_W = "Weights must be non-negative."
MSG_NZ_ZERO = (
    "The number of non-zero weights in the training set must be greater than zero."
)
MSG_NZ_DDOF = (
    "The number of non-zero weights in the training set must be greater than `ddof`."
)
MSG_NEITHER = "At least one of `return_XTX` and `return_XTY` must be True."
MSG_NO_Y = "Response variables `Y` are not provided."


def _train_std(q_t, mu, s_t, sw_t, divisor, resolution):
    """cvmatrix.py:1119-1129, same operation order."""
    var = (-2 * mu * s_t + sw_t * mu**2 + q_t) / divisor
    var = np.maximum(var, 0)
    sd = np.sqrt(var)
    return np.where(sd <= resolution, 1, sd)


class OracleCVMatrix:
    """Mirror of the reference ``CVMatrix`` (numpy backend) built on the fused restatement.

    Constructor / fit / training_* signatures follow cvmatrix.py:157-167, 207-212,
    330-332, 385-387, 451-453, 519-521.
    """

    def __init__(
        self,
        center_X: bool = True,
        center_Y: bool = True,
        scale_X: bool = True,
        scale_Y: bool = True,
        ddof: int = 1,
        dtype=np.float64,
        copy: bool = True,
    ) -> None:
        self.center_X, self.center_Y = center_X, center_Y
        self.scale_X, self.scale_Y = scale_X, scale_Y
        self.ddof = ddof
        self.dtype = dtype.type if isinstance(dtype, np.dtype) else dtype
        self.copy = copy
        self.resolution = np.finfo(dtype).resolution * 10  # cvmatrix.py:187
        self.X = self.Y = self.weights = None
        self.N = self.K = self.M = None
        self.g = None

    # ---- fit --------------------------------------------------------------------------
    def fit(self, X, Y=None, weights=None) -> None:
        self.X = _as_2d(X, self.dtype, self.copy)
        self.N, self.K = self.X.shape
        if Y is not None:
            self.Y = _as_2d(Y, self.dtype, self.copy)
            self.M = self.Y.shape[1]
        else:
            self.Y, self.M = None, None
        if weights is not None:
            self.weights = _as_2d(weights, self.dtype, self.copy)
            if bool(np.any(self.weights < 0)):  # cvmatrix.py:1188-1189
                raise ValueError(MSG_NEG_W)
        else:
            self.weights = None
        self.g = fit_globals(
            self.X, self.Y, self.weights,
            self.center_X, self.center_Y, self.scale_X, self.scale_Y,
        )
        self.XTX, self.XTY = self.g["G"], self.g["H"]
        self.sum_X, self.sum_Y = self.g["sX"], self.g["sY"]
        self.sum_sq_X, self.sum_sq_Y = self.g["qX"], self.g["qY"]
        self.sum_w, self.num_nonzero_w = self.g["sw"], self.g["nz"]

    # ---- per-fold pieces --------------------------------------------------------------
    def _train_weight_totals(self, val):
        """cvmatrix.py:589-630 (zero check only in the weighted branch, like the ref)."""
        if self.weights is None:
            t = self.dtype(self.sum_w - val.size)
            return t, t
        wv = self.weights[val]
        sw_t = self.dtype(self.sum_w - np.sum(wv))
        nz_t = self.dtype(self.num_nonzero_w - np.count_nonzero(wv))
        if nz_t == 0:
            raise ValueError(MSG_NZ_ZERO)
        return sw_t, nz_t

    def _stats(self, val, Xv, Yv, wv, mean_X, std_X, mean_Y, std_Y):
        """cvmatrix.py:632-752.  Xv/Yv are the UNWEIGHTED validation rows, wv their
        weights (None if unweighted); flags say what to return."""
        if not (mean_X or std_X or mean_Y or std_Y):
            return None, None, None, None, None
        sw_t, nz_t = self._train_weight_totals(val)
        muX = muY = sdX = sdY = None
        if mean_X or std_X:
            WXv = Xv if wv is None else Xv * wv
            sX_t = self.sum_X - WXv.sum(axis=0, keepdims=True)
            muX = sX_t / sw_t
        if mean_Y or std_Y:
            WYv = Yv if wv is None else Yv * wv
            sY_t = self.sum_Y - WYv.sum(axis=0, keepdims=True)
            muY = sY_t / sw_t
        if std_X or std_Y:
            if nz_t <= self.ddof:  # cvmatrix.py:1074-1078
                raise ValueError(MSG_NZ_DDOF)
            divisor = (nz_t - self.ddof) * sw_t / nz_t  # cvmatrix.py:1079
        if std_X:
            qX_t = self.sum_sq_X - (WXv * Xv).sum(axis=0, keepdims=True)
            sdX = _train_std(qX_t, muX, sX_t, sw_t, divisor, self.resolution)
        if std_Y:
            qY_t = self.sum_sq_Y - (WYv * Yv).sum(axis=0, keepdims=True)
            sdY = _train_std(qY_t, muY, sY_t, sw_t, divisor, self.resolution)
        return (
            muX if mean_X else None,
            sdX if std_X else None,
            muY if mean_Y else None,
            sdY if std_Y else None,
            sw_t,
        )

    @staticmethod
    def _kernel(total, WXv, Bv, muA, muB, sdA, sdB, sw_t, center):
        """cvmatrix.py:1001-1010: subtract, rank-1 centre, outer-std scale."""
        out = total - WXv.T @ Bv
        if center:
            out -= sw_t * (muA.T @ muB)
        if sdA is not None and sdB is not None:
            return out / (sdA.T @ sdB)
        if sdA is not None:
            return out / sdA.T
        if sdB is not None:
            return out / sdB
        return out

    def _training_matrices(self, rXTX: bool, rXTY: bool, val):
        """cvmatrix.py:754-896."""
        if not rXTX and not rXTY:
            raise ValueError(MSG_NEITHER)
        if rXTY and self.Y is None:
            raise ValueError(MSG_NO_Y)
        val = np.asarray(val)
        cX, cY, sX, sY = self.center_X, self.center_Y, self.scale_X, self.scale_Y
        Xv = self.X[val]
        wv = None if self.weights is None else self.weights[val]
        Yv = self.Y[val] if rXTY else None
        muX, sdX, muY, sdY, sw_t = self._stats(
            val, Xv, Yv, wv,
            mean_X=cX or (rXTY and cY),
            std_X=sX,
            mean_Y=rXTY and (cX or cY),
            std_Y=rXTY and sY,
        )
        WXv = Xv if wv is None else Xv * wv
        stats = (muX, sdX, muY, sdY)
        xtx = xty = None
        if rXTX:
            xtx = self._kernel(self.XTX, WXv, Xv, muX, muX, sdX, sdX, sw_t, cX)
        if rXTY:
            xty = self._kernel(self.XTY, WXv, Yv, muX, muY, sdX, sdY, sw_t, cX or cY)
        if rXTX and rXTY:
            return (xtx, xty), stats
        return (xtx if rXTX else xty), stats

    # ---- public API -------------------------------------------------------------------
    def training_XTX(self, validation_indices):
        return self._training_matrices(True, False, validation_indices)

    def training_XTY(self, validation_indices):
        return self._training_matrices(False, True, validation_indices)

    def training_XTX_XTY(self, validation_indices):
        return self._training_matrices(True, True, validation_indices)

    def training_statistics(self, validation_indices):
        """cvmatrix.py:519-574 (note its own flag mapping, 570-573)."""
        val = np.asarray(validation_indices)
        has_Y = self.Y is not None
        Xv = self.X[val]
        wv = None if self.weights is None else self.weights[val]
        Yv = self.Y[val] if has_Y else None
        return self._stats(
            val, Xv, Yv, wv,
            mean_X=self.center_X or self.scale_X,
            std_X=self.scale_X,
            mean_Y=(self.center_Y or self.scale_Y) and has_Y,
            std_Y=self.scale_Y and has_Y,
        )[:-1]


# --------------------------------------------------------------------------------------
# Direct ("naive") training-set computation from TRAINING indices.
# Restates tests/naive_cvmatrix.py:171-277 (the reference's own oracle); used to
# cross-check the subtract-and-correct path at sizes with no committed fixture.
# --------------------------------------------------------------------------------------
def naive_training_matrices(
    X, Y, w, train_idx, center_X, center_Y, scale_X, scale_Y, ddof,
    return_XTX=True, return_XTY=True, dtype=np.float64,
):
    X = _as_2d(X, dtype, True)
    Y = _as_2d(Y, dtype, True) if Y is not None else None
    w = _as_2d(w, dtype, True) if w is not None else None
    res = np.finfo(dtype).resolution * 10
    Xt = X[train_idx]
    wt = None if w is None else w[train_idx]
    flat_w = None if wt is None else wt.ravel()
    needs = center_X or scale_X or (return_XTY and (center_Y or scale_Y))
    if wt is not None and needs:
        nzw = dtype(np.count_nonzero(wt))
        if nzw == 0:
            raise ValueError(MSG_NZ_ZERO)
    div = None
    if wt is not None and (scale_X or (return_XTY and scale_Y)):
        if nzw <= ddof:
            raise ValueError(MSG_NZ_DDOF)
        div = (nzw - ddof) * np.sum(wt) / nzw

    def _std(mat, about):
        if wt is None:
            sd = mat.std(axis=0, ddof=ddof, keepdims=True, mean=about)
        else:
            sd = np.sqrt(np.sum(wt * (mat - about) ** 2, axis=0, keepdims=True) / div)
        sd[np.abs(sd) <= res] = 1
        return sd

    muX = sdX = muY = sdY = None
    if center_X or scale_X:
        muX = np.average(Xt, axis=0, weights=flat_w, keepdims=True)
        about = muX
        if center_X:
            Xt = Xt - muX
            about = 0
        if scale_X:
            sdX = _std(Xt, about)
            Xt = Xt / sdX
    Yt = None
    if return_XTY:
        Yt = Y[train_idx]
        if center_Y or scale_Y:
            muY = np.average(Yt, axis=0, weights=flat_w, keepdims=True)
            about = muY
            if center_Y:
                Yt = Yt - muY
                about = 0
            if scale_Y:
                sdY = _std(Yt, about)
                Yt = Yt / sdY
    XtW = Xt.T if wt is None else Xt.T * wt.T
    stats = (muX, sdX, muY, sdY)
    if return_XTX and return_XTY:
        return (XtW @ Xt, XtW @ Yt), stats
    if return_XTX:
        return XtW @ Xt, stats
    return XtW @ Yt, stats


def complement_indices(p: OraclePartitioner, fold: Hashable) -> np.ndarray:
    """Training indices of ``fold`` as the reference's tests build them
    (tests/test_cvmatrix.py:464-466): concatenation of the other folds' indices."""
    parts = [v for k, v in p.folds_dict.items() if k != fold]
    return np.concatenate(parts) if parts else np.zeros((0,), dtype=int)


# --------------------------------------------------------------------------------------
# Synthetic workload of the reference benchmark (benchmarks/benchmark.py:223-233).
# --------------------------------------------------------------------------------------
def benchmark_inputs(N: int, K: int, M: int, P: int, dtype=np.float64, seed: int = 42):
    rng = np.random.default_rng(seed=seed)
    X = rng.random((N, K), dtype=dtype)
    Y = rng.random((N, M), dtype=dtype)
    w = rng.random((N,), dtype=dtype)
    folds = np.arange(N) % P
    return X, Y, w, folds


def run_cv(X, Y, w, folds, center_X, center_Y, scale_X, scale_Y, ddof=1, dtype=np.float64):
    """One full CV pass the way benchmarks/benchmark.py:101-158 times it:
    ctor + Partitioner + fit + every fold's training_XTX_XTY.  Returns the results."""
    m = OracleCVMatrix(center_X, center_Y, scale_X, scale_Y, ddof, dtype, copy=True)
    p = OraclePartitioner(folds)
    m.fit(X, Y, w)
    return [m.training_XTX_XTY(p.get_validation_indices(f)) for f in p.folds_dict]
