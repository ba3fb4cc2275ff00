"""Improved Kernel PLS (algorithm #2 of Dayal & MacGregor 1997) on the training matrices of a
batch of folds, on the device -- the step after the cvmatrix hot path.

The reference names its consumer (README.md:23, cvmatrix/partitioner.py:27-31): the ``ikpls``
package runs this algorithm on every fold's ``(XTX, XTY)``.  Here the matrices stay where
``CVMatrix.training_XTX_XTY_batched`` wrote them (HBM) and one launch of ``cvm_pls_fit``
(include/cvmhip.h) fits all folds.  No CPU fallback."""

from __future__ import annotations

from typing import NamedTuple, Optional

import numpy as np
import torch

from . import _lib

MAX_RESPONSES = 64
MAX_COMPONENTS = 512


class PLSFit(NamedTuple):
    """``B`` (F,A,K,M): ``B[f, a]`` = regression coefficients with ``a + 1`` components;
    ``W, P, R`` (F,K,A) and ``Q`` (F,M,A) when requested, else None; ``n_fit`` (F,) int32:
    components extracted per fold (less than A only when XTY deflated to zero)."""
    B: torch.Tensor
    W: Optional[torch.Tensor]
    P: Optional[torch.Tensor]
    Q: Optional[torch.Tensor]
    R: Optional[torch.Tensor]
    n_fit: torch.Tensor


def pls_plan(n_folds: int, K: int, M: int, A: int, dtype=np.float64) -> dict:
    """How a problem is cut: row slices per fold, rows per slice, folds per launch, whether the
    slice of XTX stays in LDS (host logic only: runs without a GPU)."""
    lib = _lib.load()
    info = np.zeros(5, dtype=np.int64)
    code = _lib.CVM_F64 if np.dtype(dtype) == np.float64 else _lib.CVM_F32
    _lib.check(lib.cvm_pls_plan(n_folds, K, M, A, code, info.ctypes.data), "cvm_pls_plan")
    return {"slices": int(info[0]), "rows": int(info[1]), "folds_per_launch": int(info[2]),
            "xtx_in_lds": int(info[3]) == 1, "lds_bytes": int(info[4]),
            "kernel": "replicated small state, one barrier per component" if int(info[3]) == 2
            else ("row slices, four barriers per component" if int(info[0]) > 1 else "one workgroup per fold")}


def pls_fit_batched(XTX: torch.Tensor, XTY: torch.Tensor, A: int, *, return_factors: bool = False,
                    check: bool = True) -> PLSFit:
    """Fit A-component PLS models on ``XTX`` (F,K,K) / ``XTY`` (F,K,M) device tensors (the
    outputs of ``training_XTX_XTY_batched``; a single (K,K)/(K,M) pair is taken as F = 1).

    The routes that cut a fold into slices need an otherwise idle device (the slices of a fold wait for each
    other; ``CVM_PLS_NO_REP`` opts out of the one-barrier route).  If a slice finds no place to run, the
    barrier times out and the library recomputes every fold with one workgroup per fold in the same call;
    only where a fold does not fit one workgroup's LDS are the outputs overwritten with NaN (``n_fit`` -1) --
    ``check=True`` synchronises once to turn that case into an exception.  Nothing half-written is ever
    returned."""
    if not (isinstance(XTX, torch.Tensor) and XTX.is_cuda and isinstance(XTY, torch.Tensor) and XTY.is_cuda):
        raise TypeError("pls_fit_batched takes device tensors (the batched training matrices).")
    if XTX.dim() == 2:
        XTX = XTX.unsqueeze(0)
        XTY = XTY.unsqueeze(0) if XTY.dim() == 2 else XTY.reshape(1, -1, 1)
    if XTX.dim() != 3 or XTY.dim() != 3 or XTX.shape[1] != XTX.shape[2] or XTY.shape[:2] != XTX.shape[:2]:
        raise ValueError("XTX must be (F,K,K) and XTY (F,K,M).")
    if XTX.dtype != XTY.dtype or XTX.dtype not in (torch.float64, torch.float32):
        raise ValueError("XTX and XTY must both be float64 or both float32.")
    F, K, M = XTY.shape
    A = int(A)
    if not 1 <= A <= MAX_COMPONENTS:
        raise ValueError(f"A must be in [1, {MAX_COMPONENTS}].")
    if not 1 <= M <= MAX_RESPONSES:
        raise ValueError(f"The device PLS takes at most {MAX_RESPONSES} responses.")
    XTX = XTX.contiguous()
    XTY = XTY.contiguous()
    lib = _lib.load()
    dev = XTX.device
    code = _lib.CVM_F64 if XTX.dtype == torch.float64 else _lib.CVM_F32
    with torch.cuda.device(dev):
        nbytes = lib.cvm_pls_workspace_bytes(F, K, M, A, code)
        if nbytes == 0:
            raise ValueError("K is too large for the device PLS kernel.")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        B = torch.empty((F, A, K, M), dtype=XTX.dtype, device=dev)
        n_fit = torch.empty((F,), dtype=torch.int32, device=dev)
        status = torch.empty((1,), dtype=torch.int32, device=dev)
        W = P = Q = R = None
        if return_factors:
            W = torch.empty((F, K, A), dtype=XTX.dtype, device=dev)
            P = torch.empty_like(W)
            R = torch.empty_like(W)
            Q = torch.empty((F, M, A), dtype=XTX.dtype, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.cvm_pls_fit(_lib.ptr(XTX), _lib.ptr(XTY), F, K, M, A, code, _lib.ptr(B), _lib.ptr(W),
                             _lib.ptr(P), _lib.ptr(Q), _lib.ptr(R), _lib.ptr(n_fit), _lib.ptr(status),
                             _lib.ptr(ws), nbytes, stream)
        _lib.check(rc, "cvm_pls_fit")
        ws.record_stream(torch.cuda.current_stream(dev))
        # status 0: fine; 2: a barrier of the sliced launch timed out (a slice was not resident: another
        # stream's kernel on the device, a masked CU) and the library recomputed every fold with the kernel that
        # waits for nobody -- the outputs are valid; 1: it could not (a fold does not fit one workgroup's LDS):
        # the outputs are NaN / n_fit -1
        if check:
            st_ = int(status.item())
            pls_fit_batched.last_status = st_          # (0 or 2 after a successful call: tests, diagnostics)
            if st_ == 1:
                raise RuntimeError("cvm_pls_fit: workgroups of a fold were not co-resident (barrier timed out).")
    return PLSFit(B, W, P, Q, R, n_fit)


pls_fit_batched.last_status = None


def pls_validation_sse(cvm, folds, stats, B: torch.Tensor):
    """Weighted squared prediction errors of every fold's PLS models on the fold's own validation rows,
    on the device (``cvm_pls_validation_sse``): ``cvm`` a fitted ``CVMatrix`` with Y, ``folds`` what
    ``training_XTX_XTY_batched`` was given (a ``Partitioner``, index arrays or a ``FoldBatch``), ``stats``
    the statistics tuple that call returned, ``B`` (F,A,K,M) from ``pls_fit_batched``.  Returns
    ``(sse, wsum)``: float64 tensors (F,A,M) and (F,) -- the cross-validated RMSE with ``a + 1``
    components is ``sqrt(sse.sum(0)[a] / wsum.sum())`` (``cv_rmse``)."""
    if cvm.X is None or cvm.Y is None:
        raise ValueError("pls_validation_sse needs a CVMatrix fitted with Y.")
    batch = cvm.prepare_folds(folds)
    muX, sdX, muY, sdY = stats
    F, A, K, M = B.shape
    if (K, M) != (cvm.K, cvm.M) or F != batch.n_folds:
        raise ValueError("B does not belong to these folds / this model.")
    if cvm._Kd != cvm._Ku or (cvm._Md or 0) != (cvm._Mu or 0) or cvm.output != "torch" or cvm._out_cast:
        raise ValueError("pls_validation_sse takes device results of an unpadded float32 / float64 model "
                         "(K even, float64: M even, or copy=False).")
    # one element type and one device for everything the kernel reads: it reinterprets X, Y, the
    # weights, the statistics and B at B's element size
    dev = cvm.X.device
    if B.dtype != cvm.X.dtype or B.device != dev:
        raise ValueError(f"B is {B.dtype} on {B.device}, the model {cvm.X.dtype} on {dev}: pls_validation_sse "
                         "takes the coefficients of THIS model's training matrices.")
    for name, t, width in (("mean of X", muX, K), ("std of X", sdX, K), ("mean of Y", muY, M), ("std of Y", sdY, M)):
        if t is None:
            continue
        if t.dtype != B.dtype or t.device != dev or t.numel() != F * width:
            raise ValueError(f"the {name} is not the statistics output of these folds "
                             f"({tuple(t.shape)} {t.dtype} on {t.device}; wanted {F} x {width} {B.dtype} on {dev}).")
    lib = _lib.load()
    B = B.contiguous()
    code = _lib.CVM_F64 if B.dtype == torch.float64 else _lib.CVM_F32
    # (the longest fold, from the batch's own host offsets -- never a caller's estimate: the kernel
    #  cuts every fold into that many row chunks)
    max_rows = int(batch.sizes.max()) if batch.n_folds else 0
    with torch.cuda.device(dev):
        nbytes = lib.cvm_pls_sse_workspace_bytes(F, max_rows, M, A)
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        sse = torch.empty((F, A, M), dtype=torch.float64, device=dev)
        wsum = torch.empty((F,), dtype=torch.float64, device=dev)
        c = lambda t: None if t is None else t.contiguous()          # noqa: E731
        muX, sdX, muY, sdY = c(muX), c(sdX), c(muY), c(sdY)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.cvm_pls_validation_sse(
            _lib.ptr(cvm.X), _lib.ptr(cvm.Y), _lib.ptr(cvm.weights), _lib.ptr(batch.idx), _lib.ptr(batch.offsets), F,
            max_rows, K, M, A, code, _lib.ptr(muX), _lib.ptr(sdX), _lib.ptr(muY), _lib.ptr(sdY), _lib.ptr(B),
            _lib.ptr(sse), _lib.ptr(wsum), _lib.ptr(ws), nbytes, stream)
        _lib.check(rc, "cvm_pls_validation_sse")
        ws.record_stream(torch.cuda.current_stream(dev))
    return sse, wsum


def cv_rmse(sse: torch.Tensor, wsum: torch.Tensor) -> torch.Tensor:
    """(A, M) cross-validated RMSE per number of components and response from ``pls_validation_sse``."""
    return torch.sqrt(sse.sum(dim=0) / wsum.sum())
