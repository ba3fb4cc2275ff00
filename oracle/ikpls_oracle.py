"""
CPU oracle for the step AFTER the cvmatrix hot path: an
Improved-Kernel-PLS fit on the training matrices ``(XTX_T, XTY_T)`` of one fold --
TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s CPU leg may import it; nothing under ``cvmatrix_amd/`` does.

What it restates.  The consumer the reference names is the out-of-tree package ``ikpls``
(reference README.md:23, cvmatrix/partitioner.py:27-31): its fast cross-validation runs
"Improved Kernel PLS Algorithm #2" of Dayal & MacGregor, *Improved PLS algorithms*,
J. Chemometrics 11 (1997) 73-85, on each fold's ``XTX``/``XTY``.  ``ikpls`` is NOT present in
/root/reference (no vendored copy, no pinned version: the reference only links to it) and not
installed in the image, so this is a restatement of the PUBLISHED algorithm:

    for a = 1..A:
        w   = dominant left singular vector of XTY           (M == 1: XTY / |XTY|;
              M > 1: XTY q / |XTY q| with q the eigenvector of the largest eigenvalue of XTY^T XTY)
        r   = w - sum_{j<a} (p_j^T w) r_j
        tTt = r^T XTX r
        p   = XTX r / tTt                                    (XTX is symmetric)
        q   = XTY^T r / tTt
        XTY = XTY - (p q^T) tTt
        B_a = B_{a-1} + r q^T

Parity status: PINNED AGAINST scikit-learn, UNPINNED AGAINST ikpls ITSELF.  PLS2 by NIPALS
(``sklearn.cross_decomposition.PLSRegression``, scale=False, converged inner loop) defines the
same model: tests/test_pls_oracle.py checks the regression coefficients of this restatement
against scikit-learn 1.7.2 live and against ``tests/golden/g8_pls.npz`` (made by
``tests/golden/make_golden_pls.py``).  The sign of each component (w, p, q, r) is not
defined by the algorithm (an eigenvector's sign is arbitrary); ``B`` does not depend on it.

Stopping rule: when |XTY q| is not above ``eps`` of the dtype the remaining components cannot
be extracted; the loop stops, ``n_fit`` says how many were, and the arrays keep zeros beyond.
"""

from __future__ import annotations

import numpy as np


def ikpls_fit(XTX: np.ndarray, XTY: np.ndarray, A: int):
    """Returns ``(B, W, P, Q, R, n_fit)`` with ``B`` (A,K,M); ``W, P, R`` (K,A); ``Q`` (M,A).

    ``B[a]`` are the regression coefficients of the model with ``a + 1`` components, for
    centred/scaled predictors and responses exactly as ``XTX``/``XTY`` were.
    """
    XTX = np.asarray(XTX)
    dtype = XTX.dtype
    XTY = np.array(XTY, dtype=dtype, copy=True)
    if XTY.ndim == 1:
        XTY = XTY.reshape(-1, 1)
    K, M = XTY.shape
    B = np.zeros((A, K, M), dtype)
    W = np.zeros((K, A), dtype)
    P = np.zeros((K, A), dtype)
    Q = np.zeros((M, A), dtype)
    R = np.zeros((K, A), dtype)
    eps = np.finfo(dtype).eps
    n_fit = 0
    for a in range(A):
        if M == 1:
            w = XTY[:, 0].copy()
        else:
            S = XTY.T @ XTY
            _, vecs = np.linalg.eigh(S)
            w = XTY @ vecs[:, -1]
        nrm = np.sqrt(w @ w)
        if not nrm > eps:
            break
        w = w / nrm
        r = w.copy()
        if a:
            r -= R[:, :a] @ (P[:, :a].T @ w)
        u = XTX @ r
        tTt = r @ u
        p = u / tTt
        q = (XTY.T @ r) / tTt
        XTY -= np.outer(p, q) * tTt
        W[:, a], P[:, a], Q[:, a], R[:, a] = w, p, q, r
        B[a] = (B[a - 1] if a else 0) + np.outer(r, q)
        n_fit = a + 1
    return B, W, P, Q, R, n_fit


def predict(B: np.ndarray, Xs: np.ndarray) -> np.ndarray:
    """``Xs`` (n,K) already centred/scaled like the training data -> (A,n,M) predictions in the
    centred/scaled response space, one per number of components."""
    return np.einsum("nk,akm->anm", Xs, B)
