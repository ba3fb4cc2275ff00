"""CPU (-m "not gpu"): host-side logic of the product package -- Partitioner (incl. the
vectorised path and the CSR export), constructor / backend / dtype errors, loud failure
without a GPU, and the C-ABI library: it loads and exports every symbol include/cvmhip.h
declares (no compute calls without a GPU)."""

import ctypes
import os
import re

import numpy as np
import pytest

import cvmatrix_amd
from cvmatrix_amd import CVMatrix, Partitioner, _lib
from conftest import ROOT, load_json
from oracle.cvmatrix_oracle import OraclePartitioner


def same_partition(a, b):
    assert list(a.folds_dict) == list(b.folds_dict)
    for k in a.folds_dict:
        assert np.array_equal(a.folds_dict[k], b.folds_dict[k])
        assert a.folds_dict[k].dtype == b.folds_dict[k].dtype == np.dtype(int)


@pytest.mark.parametrize("folds", [
    np.array([0, 0, 1, 1, 2]),
    np.arange(1000) % 7,
    np.random.default_rng(0).integers(-5, 5, size=999),
    np.array([3, 3, 3]),
    np.array([True, False, True]),
    [0, "one", 2, 2],
    ["a", "b", "a", ("t", 1), ("t", 1)],
    np.array(["x", "y", "x"]),
    np.array([1.5, 2.5, 1.5]),
    [],
    np.zeros(0, dtype=int),
])
def test_partitioner_matches_reference_semantics(folds):
    """cvmatrix/partitioner.py:89-107: keys in first-seen order, ascending int indices."""
    same_partition(Partitioner(folds), OraclePartitioner(folds))


def test_partitioner_missing_fold_message():
    gold = load_json("g5_errors.json")
    for name, fn in (("fold_missing", lambda: Partitioner([0, 1, 1]).get_validation_indices(7)),
                     ("fold_missing_str", lambda: Partitioner([0, "a"]).get_validation_indices("b"))):
        with pytest.raises(ValueError) as ei:
            fn()
        assert str(ei.value) == gold[name][1]
        assert isinstance(ei.value.__cause__, KeyError)


def test_partitioner_csr_export():
    folds = np.random.default_rng(1).integers(0, 6, size=500)
    p = Partitioner(folds)
    idx, off = p.csr()
    assert idx.dtype == np.int64 and off.dtype == np.int64 and off[0] == 0 and off[-1] == 500
    for i, k in enumerate(p.folds_dict):
        assert np.array_equal(idx[off[i]:off[i + 1]], p.get_validation_indices(k))
    assert sorted(idx.tolist()) == list(range(500))


def test_constructor_surface_and_errors():
    """cvmatrix.py:157-205: flags stored, resolution = 10 * finfo.resolution; backend and
    dtype outside the device's offer are refused loudly."""
    m = CVMatrix(False, True, False, True, ddof=0, dtype=np.float32, copy=False)
    assert (m.center_X, m.center_Y, m.scale_X, m.scale_Y) == (False, True, False, True)
    assert m.ddof == 0 and m.dtype is np.float32 and m.copy is False and m.backend == "hip"
    assert m.resolution == np.finfo(np.float32).resolution * 10
    assert CVMatrix(dtype=np.dtype("float64")).dtype is np.float64
    assert CVMatrix().resolution == np.finfo(np.float64).resolution * 10  # ~1e-14
    assert m.X is None and m.XTX is None and m.sum_X is None and m.sum_w is None
    for bad in ("jax", "tpu"):
        # same form as the reference's message (cvmatrix.py:96)
        with pytest.raises(ValueError) as ei:
            CVMatrix(backend=bad)
        assert str(ei.value) == f"Invalid backend: {bad!r}. Must be 'hip' or 'numpy'."
    # the reference's default literal is accepted: the NumPy contract at the seam (ndarray results)
    mn = CVMatrix(backend="numpy")
    assert mn.backend == "numpy" and mn.output == "numpy"
    # the reference's dtype surface (tests/test_cvmatrix.py:1147-1205): float16 computes in float32,
    # wider than float64 in float64 (results as NumPy arrays of the requested type)
    mh, mw = CVMatrix(dtype=np.float16), CVMatrix(dtype=np.longdouble)
    assert mh.dtype is np.float16 and mh._npdt == np.float32 and mh.resolution == np.finfo(np.float16).resolution * 10
    assert mw.dtype is np.longdouble and mw._npdt == np.float64 and mw.output == "numpy"
    with pytest.raises(ValueError, match="Invalid output"):
        CVMatrix(output="jax")
    assert CVMatrix(output="numpy").output == "numpy" and CVMatrix().output == "torch"


def test_lazy_fit_default_follows_copy(monkeypatch):
    """fit() may defer its arithmetic only when the object owns private copies of its inputs
    (copy=True, the reference's default); copy=False aliases caller memory and computes in
    fit() like cvmatrix.py:325-328.  CVM_LAZY_FIT overrides, an explicit argument wins."""
    monkeypatch.delenv("CVM_LAZY_FIT", raising=False)
    assert CVMatrix().lazy_fit is True and CVMatrix(copy=False).lazy_fit is False
    assert CVMatrix(copy=False, lazy_fit=True).lazy_fit is True
    assert CVMatrix(copy=True, lazy_fit=False).lazy_fit is False
    monkeypatch.setenv("CVM_LAZY_FIT", "0")
    assert CVMatrix().lazy_fit is False and CVMatrix(lazy_fit=True).lazy_fit is True
    monkeypatch.setenv("CVM_LAZY_FIT", "1")
    assert CVMatrix(copy=False).lazy_fit is True


def test_library_carries_the_hash_of_its_sources():
    """cvm_source_hash() == sha256 of csrc/* + include/cvmhip.h + flags as build.py computes it:
    the binary the tests load is the committed source (the loader rebuilds or refuses otherwise)."""
    from cvmatrix_amd import build

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    lib = _lib.load()
    want = build.source_hash()
    assert lib.cvm_source_hash().decode() == want == build._embedded_hash_without_loading()
    assert want in lib.cvm_version().decode() and len(want) == 16
    for bad in (np.int32, np.complex128):
        with pytest.raises(TypeError):
            CVMatrix(dtype=bad)


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = CVMatrix()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.fit(np.ones((4, 2)))
    with pytest.raises(RuntimeError, match="fit"):
        m.training_XTX_batched([np.array([0])])
    with pytest.raises(ValueError, match="At least one of"):
        m._training_matrices(False, False, np.array([0]))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under cvmatrix_amd/ may reference it."""
    pkg = os.path.join(ROOT, "cvmatrix_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), f"{f} mentions the oracle"


def test_cabi_library_exports_every_declared_symbol():
    """include/cvmhip.h <-> libcvmhip.so <-> cvmatrix_amd/_lib.py agree on the symbol set."""
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    header = open(os.path.join(ROOT, "include", "cvmhip.h")).read()
    declared = set(re.findall(r"\b(cvm_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    loaded = _lib.load()
    assert loaded.cvm_version().decode().startswith("cvmhip")
    # host-only entry points are callable without a GPU
    assert loaded.cvm_gstats_len(512, 16) == 2 * 512 + 2 * 16 + 2
    assert loaded.cvm_fit_workspace_bytes(100000, 512, 16, _lib.CVM_F64) > 0
    assert loaded.cvm_fold_workspace_bytes(10, 100000, 10000, 512, 16, _lib.CVM_F64, 0x3F) > 0
    info = (ctypes.c_int64 * 8)()
    assert loaded.cvm_plan_fold(10, 10000, 512, 16, _lib.CVM_F64, 0x3F, 1 << 40, info) == 0
    s_off, s_diag, wgs, panels, items = info[0], info[6], info[1], info[2], info[3]
    assert panels == 4 and items == 10 and wgs == 10 * (6 * s_off + 4 * s_diag) and s_off >= 1 and s_diag >= 1
    assert info[7] == max(s_off, s_diag)
    # bad arguments are refused with a message, not a crash
    assert loaded.cvm_gram_fit(None, None, None, 1, 1, 0, 1, None, None, None, None, None, 0, None) == 1
    assert b"null pointer" in loaded.cvm_last_error()


def test_launch_planning_is_host_logic():
    """The row-split planner (cvm_plan_fold, pure host code) packs the 256 CUs by simulating the
    in-order hand-out of workgroups: off-diagonal tiles (16 MFMAs per wave and k-step) and diagonal
    tiles (11) get their own split counts -- C3's 10 folds x (6 + 4) tiles: 4 and 7, i.e. 240 long
    and 280 short workgroups instead of 500 alike in two ragged rounds; the fit stage one round;
    many mid-size folds one unit per fold (the route that finishes folds in the Gram kernel's
    epilogue)."""
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    lib = _lib.load()
    info = (ctypes.c_int64 * 8)()

    def plan(n_folds, rows, K, M, dtype=_lib.CVM_F64, flags=0x3F, ws=1 << 40):
        assert lib.cvm_plan_fold(n_folds, rows, K, M, dtype, flags, ws, info) == 0
        return dict(splits=info[7], s_off=info[0], s_diag=info[6], wgs=info[1], panels=info[2], items=info[3],
                    batch=info[4], mfma_per_4_rows=info[5])

    c3 = plan(10, 10000, 512, 16)
    assert (c3["s_off"], c3["s_diag"]) == (4, 7) and c3["wgs"] == 10 * (6 * 4 + 4 * 7) and c3["batch"] == 10
    # estimated makespan: off-diagonal items of 2500 rows first, diagonal items of 1429 rows at
    # 11/16 of the cost per row fill in behind them -- within 10 % of the 256-CU ideal
    ideal = 10 * 10000 * (6 + 4 * 11 / 16) / 256
    assert 2500 + 1429 * 11 / 16 <= 1.10 * ideal
    # executed MFMAs per 4 rows: 6 off-diagonal tiles x 64, 4 diagonal x 36 (upper triangle of the
    # tile's 8 x 8 grid of MFMA tiles), 4 x 8 for XTY (M = 16: one column tile)
    assert c3["mfma_per_4_rows"] == 6 * 64 + 4 * 36 + 4 * 8
    fit = plan(1, 100000, 512, 16, flags=0x3F | 0x80000000)
    assert fit["wgs"] <= 256 and fit["wgs"] >= 240                 # one round, nearly full
    assert fit["s_off"] > fit["s_diag"]                            # longer row ranges for the cheaper tiles
    for P in (100, 300, 1000):
        assert plan(P, 100000 // P, 512, 16)["splits"] == 1
    # a workspace that holds three folds' partials: three folds per batch, same splits or fewer
    per_fold = lib.cvm_fold_workspace_bytes(1, 10000, 10000, 512, 16, _lib.CVM_F64, 0x3F)
    small = plan(10, 10000, 512, 16, ws=int(per_fold) * 1)
    assert 1 <= small["batch"] <= 10 and small["splits"] >= 1
    # float32 with K % 4 == 0 takes the LDS-DMA kernel too (balanced diagonal tiles: 36 + 8) ...
    f32 = plan(20, 10000, 4096, 1, dtype=_lib.CVM_F32)
    assert f32["panels"] == 32 and f32["mfma_per_4_rows"] == (528 - 32) * 64 + 32 * 36 + 32 * 8
    # ... other shapes the general kernel: no skipped tiles in its count
    odd = plan(20, 10000, 4094, 1, dtype=_lib.CVM_F32)
    assert odd["mfma_per_4_rows"] == (528 - 32) * 64 + 32 * 48 + 32 * 16


def test_resident_route_rule_is_host_logic():
    """Where the round-6 resident route of the HBM regime is taken is decided on the host (csrc/host.hpp: res_shape_ok /
    res_folds_ok) and shows in the workspace the small-fold route asks for: the route keeps an operand block per fold
    there (2 x 20 rows x K floats for folds of at most 16 rows, 2 x 36 for 17 to 32).  float32, K = 2048 or a multiple of
    4096, folds of at most 32 rows under the default rule; every multiple of 1024 under mode 1; nothing under mode 0; and
    never float64."""
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    lib = _lib.load()

    def ws(K, rows, dtype=_lib.CVM_F32, folds=48):
        return int(lib.cvm_fold_workspace_bytes(folds, folds * rows, rows, K, 1, dtype, 0x3F))

    assert lib.cvm_debug_resident(3) != 0 and lib.cvm_debug_resident(-1) != 0
    try:
        assert lib.cvm_debug_resident(0) == 0
        base = {K: ws(K, 16) for K in (1024, 2048, 3072, 4096, 8192)}
        base32 = ws(4096, 32)
        assert lib.cvm_debug_resident(2) == 0
        for K in (2048, 4096, 8192):
            assert ws(K, 16) > base[K] and ws(K, 16) >= 48 * 2 * 20 * K * 4, K     # (the workspace is a maximum over routes)
        for K in (1024, 3072):
            assert ws(K, 16) == base[K], K
        assert ws(4096, 32) > base32 and ws(4096, 32) >= 48 * 2 * 36 * 4096 * 4
        assert ws(4096, 16, _lib.CVM_F64) == int(lib.cvm_fold_workspace_bytes(48, 48 * 16, 16, 4096, 1, _lib.CVM_F64, 0x3F))
        assert lib.cvm_debug_resident(1) == 0
        for K in (1024, 3072):
            assert ws(K, 16) > base[K] and ws(K, 16) >= 48 * 2 * 20 * K * 4, K
        assert ws(1000, 16) == int(lib.cvm_fold_workspace_bytes(48, 48 * 16, 16, 1000, 1, _lib.CVM_F32, 0x3F))
    finally:
        lib.cvm_debug_resident(2)


def test_package_metadata():
    assert cvmatrix_amd.__all__ == ["CVMatrix", "Partitioner", "FoldBatch"]


def test_header_constants_match_the_python_binding():
    """Status codes, dtype codes and flag bits of include/cvmhip.h as cvmatrix_amd/_lib.py uses them."""
    header = open(os.path.join(ROOT, "include", "cvmhip.h")).read()
    defs = {k: int(v.rstrip("u"), 0) for k, v in re.findall(r"#define\s+(CVM_[A-Z0-9_]+)\s+(0x[0-9A-Fa-f]+u?|\d+u?)\b", header)}
    expect = {
        "CVM_OK": _lib.CVM_OK, "CVM_EINVAL": _lib.CVM_EINVAL, "CVM_EWORKSPACE": _lib.CVM_EWORKSPACE,
        "CVM_ELAUNCH": _lib.CVM_ELAUNCH, "CVM_F32": _lib.CVM_F32, "CVM_F64": _lib.CVM_F64,
        "CVM_RET_XTX": _lib.RET_XTX, "CVM_RET_XTY": _lib.RET_XTY, "CVM_CENTER_X": _lib.CENTER_X,
        "CVM_CENTER_Y": _lib.CENTER_Y, "CVM_SCALE_X": _lib.SCALE_X, "CVM_SCALE_Y": _lib.SCALE_Y,
        "CVM_IDX_HOST": _lib.IDX_HOST,
    }
    for name, value in expect.items():
        assert defs.get(name) == value, (name, defs.get(name), value)


def test_exact_index_comparison_helpers():
    """The comparisons behind every served loop (cvmatrix_amd/cvmatrix.py): exact, not sampled."""
    from cvmatrix_amd.cvmatrix import _same_indices, _same_objects, _NO_WEIGHTS, _WeightsToken

    a = np.arange(0, 100000, 10)
    b = a.copy()
    assert _same_indices(a, b)
    b[4321] += 1                                   # one interior element, not an end, not on any stride
    assert not _same_indices(a, b)
    b = a.copy(); b[100] += 5; b[200] -= 5          # same size, ends and sum
    assert b.sum() == a.sum() and not _same_indices(a, b)
    assert not _same_indices(a, a[:-1]) and _same_indices(a[::2], a[::2].copy())      # (non-contiguous views too)
    assert _same_indices(a.astype(np.int32), a)                                       # other dtype, same values
    x, y = np.arange(3), np.arange(3)
    assert _same_objects([x, y], [x, y]) and not _same_objects([x, y], [x, y.copy()]) and not _same_objects([x], [x, y])
    assert _WeightsToken() is not _WeightsToken() and _NO_WEIGHTS is _NO_WEIGHTS


def test_local_counts_decide_the_checks_without_the_global_ones():
    """CVMatrix._passes_on_local_counts: this process's counts are lower bounds of the global ones; when
    even they leave every training set more non-zero weights than ddof no raise is possible."""
    from cvmatrix_amd.cvmatrix import FoldBatch

    m = CVMatrix(ddof=1)
    sizes = np.array([10, 10], dtype=np.int64)
    fb = FoldBatch(None, None, np.array([0, 10, 20], dtype=np.int64), sizes.copy(), None, np.arange(20), 20)
    m.weights, m._n_total, m._nz_total = None, 20, 20
    assert m._passes_on_local_counts(fb, True, True)              # 20 - 10 = 10 > ddof
    one = FoldBatch(None, None, np.array([0, 20], dtype=np.int64), np.array([20]), None, np.arange(20), 20)
    assert not m._passes_on_local_counts(one, True, True)         # the rank's only fold is all of its rows: must ask
    assert m._passes_on_local_counts(one, False, False)           # nothing to check
    m.weights, m._nz_total = object(), 12                         # weighted: counts of non-zero weights
    fb.nz_val = np.array([10, 11])
    assert not m._passes_on_local_counts(fb, True, True) and m._passes_on_local_counts(fb, True, True, only=0)
    m.ddof = 0
    assert m._passes_on_local_counts(fb, True, True)


def test_emulated_rank_is_importable_without_a_gpu():
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from emulate import EmulatedRank      # (tools/emulate.py: a bench helper, not part of the package)

    r = EmulatedRank(emu_world=8, emu_rank=3, lazy_fit=True)
    assert r.world == 8 and r.rank == 3 and r._exchanges_globals() and r.lazy_fit


def test_periodic_labels_take_the_formula_and_build_the_same_dict():
    """Partitioner._init_periodic (arange(N) % P, arange(N)): the same keys, order, dtypes and index
    arrays as the sorting path, for periodic, nearly periodic and aperiodic labels."""
    import cvmatrix_amd.partitioner as pm
    from cvmatrix_amd import Partitioner

    def general(arr):
        orig = pm.Partitioner._init_periodic
        pm.Partitioner._init_periodic = lambda self, a: False
        try:
            return Partitioner(arr)
        finally:
            pm.Partitioner._init_periodic = orig

    rng = np.random.default_rng(0)
    cases = [np.arange(1000) % 10, np.arange(1003) % 7, np.arange(50), np.zeros(9, dtype=np.int64), np.array([5]),
             (np.arange(40) % 4) * 3 + 2, np.r_[np.arange(30) % 5, 7], rng.integers(0, 4, 200),
             np.array([3, 1, 2, 3, 1, 2, 3, 1]), np.arange(20)[::-1].copy(), np.arange(64) % 2 == 0,
             np.arange(12, dtype=np.int8) % 3, np.array([0, 1, 0, 1, 1, 0]), np.arange(9000) % 4500]
    took = 0
    for a in cases:
        p, q = Partitioner(a), general(a)
        took += bool(pm.Partitioner._init_periodic(Partitioner.__new__(Partitioner), a)) if a.size else 0
        assert list(p.folds_dict.keys()) == list(q.folds_dict.keys())
        for k in p.folds_dict:
            x, y = p.folds_dict[k], q.folds_dict[k]
            assert x.dtype == y.dtype and np.array_equal(x, y)
        for i, k in enumerate(p.folds_dict):
            if i < 3:
                assert pm.partitioner_pos(p.folds_dict[k]) == (p, i)
        with pytest.raises(ValueError, match="Fold nope not found."):
            p.get_validation_indices("nope")
    assert took >= 8          # the periodic ones did take the formula


def test_plain_bench_command_with_several_gpus_launches_its_own_ranks(monkeypatch):
    """`python3 bench.py --gpus N` (no launcher, the form the driver uses at N = 1): the parent builds a
    `torch.distributed.run` command with one rank per GPU on a free loopback port and returns the child's exit code,
    before torch is imported in the parent (nothing there may touch the GPU)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-5] == os.path.join(root, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
