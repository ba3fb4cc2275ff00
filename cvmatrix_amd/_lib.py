"""ctypes binding of libcvmhip.so (C ABI: include/cvmhip.h).

torch is imported first on purpose: PyTorch-ROCm ships its own libamdhip64.so.7 and the
kernels must run in that HIP runtime instance so that torch's device pointers and streams
are valid inside the library (same SONAME -> the loader reuses the loaded runtime)."""

from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime that libcvmhip.so binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
# CVM_LIB_PATH: load another build of the library (experimental builds under tools/; implies
# CVM_SKIP_HASH_CHECK)
LIB_PATH = os.environ.get("CVM_LIB_PATH") or os.path.join(_HERE, "libcvmhip.so")

CVM_OK, CVM_EINVAL, CVM_EWORKSPACE, CVM_ELAUNCH = 0, 1, 2, 3
CVM_F32, CVM_F64 = 0, 1
RET_XTX, RET_XTY = 0x01, 0x02
CENTER_X, CENTER_Y, SCALE_X, SCALE_Y = 0x04, 0x08, 0x10, 0x20
IDX_HOST = 0x40

EXPORTS = (
    "cvm_version", "cvm_source_hash", "cvm_last_error", "cvm_gstats_len", "cvm_fit_workspace_bytes",
    "cvm_gram_fit", "cvm_fold_workspace_bytes", "cvm_fold_update", "cvm_fold_update_ex", "cvm_plan_fold", "cvm_debug_force_splits", "cvm_debug_resident",
    "cvm_timing_enable", "cvm_timing_read", "cvm_timing_read_kinds", "cvm_fill_probe", "cvm_clock_probe",
    "cvm_sweep_workspace_bytes", "cvm_sweep_fit", "cvm_sweep_folds", "cvm_sweep_fold_range", "cvm_sweep_all",
    "cvm_partition_workspace_bytes", "cvm_partition_labels", "cvm_partition_periodic", "cvm_weights_check",
    "cvm_pls_workspace_bytes", "cvm_pls_fit", "cvm_pls_plan",
    "cvm_pls_sse_workspace_bytes", "cvm_pls_validation_sse",
)

_lib = None


def load():
    """Load the shared library once; fail loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -m cvmatrix_amd.build, or __graft_entry__.build()). "
            "cvmatrix_amd has no CPU fallback."
        )
    # the binary must be the source next to it: a stale prebuilt library is rebuilt when a
    # compiler is there, refused otherwise (CVM_SKIP_HASH_CHECK=1: use it as it is)
    from . import build as _build

    if (os.environ.get("CVM_SKIP_HASH_CHECK", "0") == "0" and not os.environ.get("CVM_LIB_PATH")
            and os.path.isdir(os.path.join(_HERE, "csrc"))):
        want, have = _build.source_hash(), _build._embedded_hash_without_loading(LIB_PATH)
        if have != want:
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            if not os.path.exists(hipcc):
                raise ImportError(
                    f"{LIB_PATH} was built from other sources (library {have}, sources {want}) "
                    "and there is no hipcc to rebuild it")
            print(f"cvmatrix_amd: libcvmhip.so is stale (library {have}, sources {want}); rebuilding",
                  flush=True)
            _build.build(force=False, verbose=False)   # (build() re-checks the hash under its lock: one builds, the others find it current)
    lib = C.CDLL(LIB_PATH)
    vp, i64, sz, u32, dbl = C.c_void_p, C.c_int64, C.c_size_t, C.c_uint, C.c_double
    lib.cvm_version.restype = C.c_char_p
    lib.cvm_source_hash.restype = C.c_char_p
    lib.cvm_last_error.restype = C.c_char_p
    lib.cvm_gstats_len.restype = sz
    lib.cvm_gstats_len.argtypes = [C.c_int, C.c_int]
    lib.cvm_fit_workspace_bytes.restype = sz
    lib.cvm_fit_workspace_bytes.argtypes = [i64, C.c_int, C.c_int, C.c_int]
    lib.cvm_gram_fit.restype = C.c_int
    lib.cvm_gram_fit.argtypes = [vp, vp, vp, i64, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp,
                                 vp, sz, vp]
    lib.cvm_fold_workspace_bytes.restype = sz
    lib.cvm_fold_workspace_bytes.argtypes = [i64, i64, i64, C.c_int, C.c_int, C.c_int, u32]
    lib.cvm_fold_update.restype = C.c_int
    lib.cvm_fold_update.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, C.c_int, C.c_int,
                                    C.c_int, u32, dbl, dbl, vp, vp, vp, vp, vp, vp, vp, vp,
                                    vp, vp, vp, sz, vp]
    lib.cvm_fold_update_ex.restype = C.c_int
    lib.cvm_fold_update_ex.argtypes = lib.cvm_fold_update.argtypes + [vp]
    lib.cvm_sweep_workspace_bytes.restype = sz
    lib.cvm_sweep_workspace_bytes.argtypes = [i64, i64, C.c_int, C.c_int, C.c_int]
    lib.cvm_sweep_fit.restype = C.c_int
    lib.cvm_sweep_fit.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, C.c_int, C.c_int, C.c_int,
                                  vp, vp, vp, vp, vp, sz, vp, vp]
    lib.cvm_sweep_all.restype = C.c_int
    lib.cvm_sweep_all.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, C.c_int, C.c_int, C.c_int, u32, dbl, dbl,
                                  vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp]
    lib.cvm_sweep_folds.restype = C.c_int
    lib.cvm_sweep_folds.argtypes = [vp, i64, C.c_int, C.c_int, C.c_int, u32, dbl, dbl, C.c_int,
                                    vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i64, vp]
    lib.cvm_sweep_fold_range.restype = C.c_int
    lib.cvm_sweep_fold_range.argtypes = [vp, i64, i64, i64, C.c_int, C.c_int, C.c_int, u32, dbl, dbl, C.c_int,
                                         vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i64, vp]
    lib.cvm_partition_workspace_bytes.restype = sz
    lib.cvm_partition_workspace_bytes.argtypes = [i64, i64]
    lib.cvm_partition_labels.restype = C.c_int
    lib.cvm_partition_labels.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, sz, vp]
    lib.cvm_partition_periodic.restype = C.c_int
    lib.cvm_partition_periodic.argtypes = [vp, i64, i64, vp, C.c_int, vp, vp, vp, vp, vp]
    lib.cvm_weights_check.restype = C.c_int
    lib.cvm_weights_check.argtypes = [vp, i64, C.c_int, vp, vp]
    lib.cvm_pls_workspace_bytes.restype = sz
    lib.cvm_pls_workspace_bytes.argtypes = [i64, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.cvm_pls_fit.restype = C.c_int
    lib.cvm_pls_fit.argtypes = [vp, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp,
                                vp, vp, vp, sz, vp]
    lib.cvm_pls_plan.restype = C.c_int
    lib.cvm_pls_plan.argtypes = [i64, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.cvm_pls_sse_workspace_bytes.restype = sz
    lib.cvm_pls_sse_workspace_bytes.argtypes = [i64, i64, C.c_int, C.c_int]
    lib.cvm_pls_validation_sse.restype = C.c_int
    lib.cvm_pls_validation_sse.argtypes = [vp, vp, vp, vp, vp, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int,
                                           vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.cvm_timing_enable.restype = C.c_int
    lib.cvm_timing_enable.argtypes = [C.c_int]
    lib.cvm_timing_read.restype = C.c_int
    lib.cvm_timing_read.argtypes = [vp, vp, vp, vp]
    lib.cvm_timing_read_kinds.restype = C.c_int
    lib.cvm_timing_read_kinds.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.cvm_fill_probe.restype = C.c_int
    lib.cvm_fill_probe.argtypes = [vp, sz, vp]
    lib.cvm_clock_probe.restype = C.c_int
    lib.cvm_clock_probe.argtypes = [vp, sz]
    lib.cvm_debug_force_splits.restype = C.c_int
    lib.cvm_debug_force_splits.argtypes = [C.c_int, C.c_int]
    lib.cvm_debug_resident.restype = C.c_int
    lib.cvm_debug_resident.argtypes = [C.c_int]
    lib.cvm_plan_fold.restype = C.c_int
    lib.cvm_plan_fold.argtypes = [i64, i64, C.c_int, C.c_int, C.c_int, u32, sz, vp]
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != CVM_OK:
        msg = load().cvm_last_error().decode()
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr(t) -> int:
    """Device (or host) address of a tensor / ndarray, 0 for None."""
    if t is None:
        return 0
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data
