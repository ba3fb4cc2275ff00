"""GPU: the drop-in boundary -- NumPy-returning mode (the reference's README loop and example
run with only the import changed), honest fit() defaults, input validation that cannot be
fooled by recycled device addresses, prepared fold batches that outlive a refit."""

import os

import numpy as np
import pytest

from conftest import assert_normwise, assert_stats, to_np
from oracle.cvmatrix_oracle import OracleCVMatrix, OraclePartitioner

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd(hip_device):
    import cvmatrix_amd
    from cvmatrix_amd import _lib

    _lib.load()
    return cvmatrix_amd


def test_readme_quick_start_runs_with_numpy_results(amd):
    """The call sequence of the reference's quick start (README.md:96-141: N=100, K=50, M=10,
    5 folds arange(100) % 5, weights U(0,1)+0.1, all flags on; per fold training_XTX_XTY, then
    training_XTX, then training_XTY) with output="numpy": every result is an ndarray of the
    reference's shape and dtype and equals the oracle."""
    CVMatrix, Partitioner = amd.CVMatrix, amd.Partitioner
    rng = np.random.default_rng(2024)
    N, K, M = 100, 50, 10
    X = rng.uniform(size=(N, K))
    Y = rng.uniform(size=(N, M))
    folds = np.arange(100) % 5
    weights = rng.uniform(size=(N,)) + 0.1
    cvm = CVMatrix(center_X=True, center_Y=True, scale_X=True, scale_Y=True, output="numpy")
    cvm.fit(X=X, Y=Y, weights=weights)
    ref = OracleCVMatrix()
    ref.fit(X, Y, weights)
    p = Partitioner(folds=folds)
    for fold in p.folds_dict:
        val_indices = p.get_validation_indices(fold)
        result = cvm.training_XTX_XTY(val_indices)
        (XTWX, XTWY) = result[0]
        stats = result[1]
        assert all(type(a) is np.ndarray and a.dtype == np.float64 for a in (XTWX, XTWY, *stats))
        assert XTWX.shape == (K, K) and XTWY.shape == (K, M)
        assert stats[0].shape == stats[1].shape == (1, K) and stats[2].shape == stats[3].shape == (1, M)
        (rx, ry), rst = ref.training_XTX_XTY(val_indices)
        assert_normwise(XTWX, rx, 1e-10, "readme XTX")
        assert_normwise(XTWY, ry, 1e-10, "readme XTY")
        assert_stats(stats, rst, 1e-10, "readme stats")
        # NumPy on the results, as the reference's callers do (tests/test_cvmatrix.py:489-490)
        assert np.allclose(XTWX, XTWX.T) and np.isfinite(XTWY / stats[1].T).all()
        x_only, st_x = cvm.training_XTX(val_indices)
        assert type(x_only) is np.ndarray and st_x[2] is None and st_x[3] is None
        np.testing.assert_allclose(x_only, XTWX, rtol=0, atol=1e-9)
        y_only, st_y = cvm.training_XTY(val_indices)
        assert type(y_only) is np.ndarray and all(s is not None for s in st_y)
        np.testing.assert_allclose(y_only, XTWY, rtol=0, atol=1e-9)
        st_only = cvm.training_statistics(val_indices)
        assert all(type(s) is np.ndarray for s in st_only)
    # the fitted attributes are ndarrays too (cvmatrix.py:1215-1241)
    for a, r in ((cvm.XTX, ref.XTX), (cvm.XTY, ref.XTY), (cvm.sum_X, ref.sum_X), (cvm.sum_sq_X, ref.sum_sq_X),
                 (cvm.sum_Y, ref.sum_Y), (cvm.sum_sq_Y, ref.sum_sq_Y)):
        assert type(a) is np.ndarray
        assert_normwise(a, r, 1e-12, "attribute")
    assert cvm.XTX is cvm.XTX                       # copied once per fit
    # batched calls follow the same switch
    (bx, by), bst = cvm.training_XTX_XTY_batched(p)
    assert type(bx) is np.ndarray and bx.shape == (5, K, K) and bst[0].shape == (5, 1, K)


def test_default_fit_is_eager_for_aliased_inputs_and_lazy_for_private_copies(amd, hip_device, monkeypatch):
    import torch

    monkeypatch.delenv("CVM_LAZY_FIT", raising=False)
    rng = np.random.default_rng(3)
    X = torch.from_numpy(rng.random((600, 40))).to(hip_device)
    Y = torch.from_numpy(rng.random((600, 3))).to(hip_device)
    eager = amd.CVMatrix(copy=False)
    eager.fit(X, Y)
    assert not eager._pending and eager.X.data_ptr() == X.data_ptr()
    G = eager.XTX.clone()
    lazy = amd.CVMatrix()                       # copy=True: private copies, deferral is safe
    lazy.fit(X, Y)
    assert lazy._pending and lazy.X.data_ptr() != X.data_ptr()
    X.mul_(2.0)                                 # the caller scribbles over its array after fit()
    assert torch.equal(lazy.XTX, G)             # the lazy object computes from its own copy
    assert torch.equal(eager.XTX, G)            # the eager one computed inside fit()


@pytest.mark.serving
@pytest.mark.parametrize("mode", ["sync", "deferred"])
def test_weights_are_validated_again_when_a_new_tensor_reuses_the_address(amd, hip_device, mode):
    """the device-weights validation cache must not be fooled by a recycled address.  Both validation
    modes: ``sync`` raises inside fit() (a read-back), ``deferred`` (the default) at the first hand-out of that
    fit (counts taken on the device, nothing read back)."""
    import torch

    def refused(m, *a, **kw):
        """fit + first use: the reference's raise, from fit (sync) or from the first result (deferred)"""
        with pytest.raises(ValueError, match="Weights must be non-negative."):
            m.fit(*a, **kw)
            assert mode == "deferred"                      # (sync never gets here)
            m.training_XTX(np.arange(10))

    rng = np.random.default_rng(4)
    N = 5000
    X = torch.from_numpy(rng.random((N, 16))).to(hip_device)
    m = amd.CVMatrix(copy=True, validate_weights=mode)
    hit = False
    for _ in range(20):
        w = torch.rand(N, dtype=torch.float64, device=hip_device)
        addr = w.data_ptr()
        m.fit(X, None, w)
        m.training_XTX(np.arange(10))
        del w
        w2 = torch.rand(N, dtype=torch.float64, device=hip_device)
        w2[7] = -1.0
        w2 = w2.clone() if w2.data_ptr() != addr else w2      # whatever address it got
        hit = hit or (w2.data_ptr() == addr)
        refused(m, X, None, w2)
        del w2
    # by default every fit checks its inputs again, like the reference (cvmatrix.py:207-328) ...
    w = torch.rand(N, dtype=torch.float64, device=hip_device)
    m = amd.CVMatrix(copy=True, serve_loops=True, validate_weights=mode)
    m.fit(X, None, w)
    m.training_XTX(np.arange(10))
    if mode == "sync":
        host = m._w_host
        m.fit(X, None, w)
        assert m._w_host is not host
        # ... a caller who says the tensors are unchanged is believed as far as torch's version counters agree
        # (no second read-back of the same unmodified tensor object) ...
        host = m._w_host
        m.fit(X, None, w, assume_unchanged=True)
        assert m._w_host is host
    else:
        # (a call with a handful of rows needs the exact count of non-zero weights among them and reads the weights
        #  back for it; folds that leave plenty of rows for training never do)
        assert m._w_verified
        m2 = amd.CVMatrix(copy=True, validate_weights=mode)
        m2.fit(X, None, w)
        m2.training_XTX_batched([np.arange(100), np.arange(100, 300)])
        assert m2._w_host is None and m2._w_verified       # counted on the device, never read back
        gen = m._w_gen
        m.fit(X, None, w)
        assert m._wchk and m._w_gen is not gen             # checked again (a new check is on its way)
        m.training_XTX(np.arange(10))
        gen = m._w_gen
        m.fit(X, None, w, assume_unchanged=True)
        assert not m._wchk and m._w_gen is gen             # believed: no new check
    # ... until the tensor is modified in place
    w[3] = -2.0
    refused(m, X, None, w, assume_unchanged=True)
    if mode == "deferred":
        # a fit whose weights failed keeps raising at every hand-out, attributes included, until the next fit
        with pytest.raises(ValueError, match="Weights must be non-negative."):
            m.XTX
        with pytest.raises(ValueError, match="Weights must be non-negative."):
            m.training_statistics(np.arange(10))
        w[3] = 0.5
        m.fit(X, None, w)
        m.training_XTX(np.arange(10))


def test_writes_behind_torchs_back_are_seen_unless_the_caller_vouches_for_the_tensors(amd, hip_device):
    """The blind spot of the fit() fast path, and who carries it.  torch's version counter does not see a
    write through ``t.data`` (nor DLPack consumers, raw-pointer kernels, other libraries).  DEFAULT: fit()
    re-reads its inputs on every call like the reference (cvmatrix.py:207-328) -- new X, new weights and a
    negative weight smuggled in that way are all seen.  ``trust_tensor_versions=True`` /
    ``fit(assume_unchanged=True)``: the caller vouches for the tensors, and a trusting fit of private copies
    (``copy=True``) returns the PREVIOUS data's matrices -- the documented price of the 30 us it saves."""
    import torch

    rng = np.random.default_rng(8)
    N, K = 4000, 24
    X = torch.from_numpy(rng.random((N, K))).to(hip_device)
    w = torch.from_numpy(rng.random(N) + 0.1).to(hip_device)
    for lazy in (False, True):
        m = amd.CVMatrix(copy=True, lazy_fit=lazy)                     # the default policy
        m.fit(X, None, w)
        g0 = m.XTX.clone()
        X.data.mul_(2.0)                                               # no version bump
        m.fit(X, None, w)
        assert_normwise(m.XTX, 4.0 * to_np(g0), 1e-12, "refit after a .data write")
        X.data.mul_(0.5)
        w.data[5] = -1.0
        with pytest.raises(ValueError, match="Weights must be non-negative."):
            m.fit(X, None, w)                                          # (deferred validation: raised by the first use)
            m.XTX
        w.data[5] = 0.5
        t = amd.CVMatrix(copy=True, lazy_fit=lazy, trust_tensor_versions=True, serve_loops=True)
        t.fit(X, None, w)
        g1 = t.XTX.clone()
        X.data.mul_(2.0)
        t.fit(X, None, w)                                              # believed: still the old private copy
        assert torch.equal(t.XTX, g1)
        t.fit(X, None, w, assume_unchanged=False)                      # the caller knows better this time
        assert_normwise(t.XTX, 4.0 * to_np(g1), 1e-12, "assume_unchanged=False")
        X.data.mul_(0.5)


def test_fold_batch_is_checked_against_the_fit_it_is_used_with(amd):
    """a prepared FoldBatch used after a refit -- other row count: refused; other
    weights: its non-zero counts are recomputed, so the reference's raises stay right."""
    rng = np.random.default_rng(5)
    X, Y = rng.random((400, 12)), rng.random((400, 2))
    w = rng.random(400)
    m = amd.CVMatrix(ddof=1)
    m.fit(X, Y, w)
    folds = [np.arange(0, 200), np.arange(200, 400)]
    batch = m.prepare_folds(folds)
    m.training_XTX_XTY_batched(batch)
    m.fit(X[:300], Y[:300], w[:300])
    with pytest.raises(ValueError, match="prepared for 400 samples"):
        m.training_XTX_XTY_batched(batch)
    # refit with weights that are zero outside fold 0: fold 1's training set = fold 0 keeps weights,
    # fold 0's training set has none -> the reference's message (cvmatrix.py:626-629)
    w2 = w.copy()
    w2[200:] = 0.0
    m.fit(X, Y, w2)
    with pytest.raises(ValueError, match="greater than zero"):
        m.training_XTX_XTY_batched(batch)
    ref = OracleCVMatrix()
    ref.fit(X, Y, w2)
    with pytest.raises(ValueError, match="greater than zero"):
        ref.training_XTX_XTY(folds[0])
    # and back: the same batch is fine again with the first weights
    m.fit(X, Y, w)
    (bx, by), st = m.training_XTX_XTY_batched(batch)
    ref.fit(X, Y, w)
    (rx, ry), rst = ref.training_XTX_XTY(folds[1])
    assert_normwise(bx[1], rx, 1e-10, "batch reuse")


def test_fit_that_raises_leaves_no_pending_state(amd):
    rng = np.random.default_rng(6)
    m = amd.CVMatrix(lazy_fit=True)
    m.fit(rng.random((300, 8)), rng.random((300, 2)))
    assert m._pending
    with pytest.raises(ValueError):
        m.fit(rng.random((100, 8)), rng.random((90, 2)))          # row mismatch, raised after X is stored
    assert not m._pending and m._sweep is None


@pytest.mark.serving
def test_reference_loop_is_served_from_one_sweep(amd, hip_device):
    """The reference's loop -- fit, then training_XTX_XTY(p.get_validation_indices(fold)) per fold --
    with the default (lazy, private copies) object: the first call recognises the Partitioner's
    own index array, sweeps all folds once, and every call returns the batched path's bits.  Other
    index arrays (copies, other folds, a second Partitioner) take the ordinary route and agree to
    rounding; statistics-only and XTX-only / XTY-only calls work from the same sweep."""
    import torch

    rng = np.random.default_rng(11)
    N, K, M, P = 6000, 96, 4, 6
    X, Y, w = rng.random((N, K)), rng.random((N, M)), rng.random(N) + 0.01
    p = amd.Partitioner(rng.integers(0, P, size=N))
    ref = amd.CVMatrix(lazy_fit=False)
    ref.fit(X, Y, w)
    (bx, by), bst = ref.training_XTX_XTY_batched(p)
    m = amd.CVMatrix()                               # copy=True -> lazy
    m.fit(X, Y, w)
    assert m._pending
    keys = list(p.folds_dict)
    for i, k in enumerate(keys):
        (xtx, xty), st = m.training_XTX_XTY(p.get_validation_indices(k))
        if i == 0:
            assert not m._pending and m._sweep is not None and m._sweep_ids is not None
        assert_normwise(xtx, bx[i].cpu().numpy(), 1e-11, "loop XTX")
        assert_normwise(xty, by[i].cpu().numpy(), 1e-11, "loop XTY")
        for a, b in zip(st, bst):
            np.testing.assert_allclose(a.cpu().numpy(), b[i].cpu().numpy(), rtol=1e-11)
        xtx.zero_(); xty.zero_()                     # the caller owns what it was handed ...
    assert m._sweep_cache is None                    # (every fold's slice was handed out once)
    # ... a second request for a fold is computed again: bits of the batched sweep path
    m2 = amd.CVMatrix()
    m2.fit(X, Y, w)
    (sx, sy), sst = m2.training_XTX_XTY_batched(p)
    assert m2._sweep is not None
    for i, k in enumerate(keys):
        (xtx, xty), st = m.training_XTX_XTY(p.get_validation_indices(k))
        assert torch.equal(xtx, sx[i]) and torch.equal(xty, sy[i])
        x_only, st_x = m.training_XTX(p.get_validation_indices(k))
        y_only, st_y = m.training_XTY(p.get_validation_indices(k))
        assert torch.equal(x_only, sx[i]) and torch.equal(y_only, sy[i])
        assert st_x[2] is None and st_x[3] is None and all(s is not None for s in st_y)
    # a copy of the indices is not recognised: ordinary route, same numbers to rounding
    v = p.get_validation_indices(keys[2]).copy()
    (xtx, xty), _ = m.training_XTX_XTY(v)
    assert_normwise(xtx, sx[2].cpu().numpy(), 1e-11, "copy of the indices")
    # an array changed in place after the sweep is not served from it
    v2 = p.get_validation_indices(keys[1])
    saved = v2.copy()
    v2[0], v2[-1] = v2[-1], v2[0]
    (xtx, _), _ = m.training_XTX_XTY(v2)
    assert_normwise(xtx, sx[1].cpu().numpy(), 1e-11, "permuted fold")
    v2[:] = saved
    assert m._sweep_fold_of(v2) == 1
    other = int(p.get_validation_indices(keys[0])[5])
    v2[v2.size // 2] = other                          # a row of another fold, in the middle
    assert m._sweep_fold_of(v2) is None               # (exact comparison: any change is seen)
    v2[:] = saved
    # folds that do not partition the rows: no sweep, the fit kernel runs
    q = amd.Partitioner(np.arange(N) % 5)
    m3 = amd.CVMatrix()
    m3.fit(X, Y, w)
    sub = q.get_validation_indices(0)[:500].copy()
    m3.training_XTX_XTY(sub)
    assert m3._sweep is None and not m3._pending


@pytest.mark.serving
@pytest.mark.parametrize("N,K,M,labels_kind,dtype", [(700, 50, 3, "loo", np.float64), (6000, 36, 2, "mod150", np.float64),
                                                     (5000, 40, 0, "random400", np.float64), (900, 64, 1, "loo", np.float32)])
def test_per_fold_loop_over_many_folds_is_read_ahead(amd, N, K, M, labels_kind, dtype):
    """The reference's loop ``for fold in p.folds_dict: cvm.training_XTX_XTY(p.get_validation_indices(fold))``
    over a Partitioner with MANY folds (leave-one-out, benchmarks/benchmark.py:153-158): the calls are
    recognised by the identity of the Partitioner's arrays and served from chunks computed by one
    batched launch sequence each -- same bits as the batched call, the reference's raises still at the
    offending call, arrays changed in place (and copies) take the ordinary route."""
    import torch

    rng = np.random.default_rng(N + K)
    X = rng.random((N, K)).astype(dtype)
    Y = rng.random((N, M)).astype(dtype) if M else None
    w = rng.random(N).astype(dtype)
    w[rng.choice(N, N // 7, replace=False)] = 0
    labels = {"loo": np.arange(N), "mod150": np.arange(N) % 150, "random400": rng.integers(0, 400, N)}[labels_kind]
    p = amd.Partitioner(labels)
    keys = list(p.folds_dict)
    ref = amd.CVMatrix(dtype=dtype, lazy_fit=False)
    ref.fit(X, Y, w)
    if M:
        (bx, by), bst = ref.training_XTX_XTY_batched(p)
    else:
        bx, bst = ref.training_XTX_batched(p)
    m = amd.CVMatrix(dtype=dtype)
    m.fit(X, Y, w)
    served = 0
    # folds of at most 32 rows go through kernels without row splits: the same bits however the folds
    # are batched; larger folds get the row-split plan of their batch (a chunk here, all folds there)
    exact = labels_kind == "loo"
    tol = 1e-12 if dtype is np.float64 else 1e-5

    def same(a, b, what):
        if exact:
            assert torch.equal(a, b), what
        else:
            assert_normwise(a.double(), b.double().cpu().numpy(), tol, what)

    for i, k in enumerate(keys):
        v = p.get_validation_indices(k)
        if M:
            (xtx, xty), st = m.training_XTX_XTY(v)
            same(xty, by[i], "XTY")
        else:
            xtx, st = m.training_XTX(v)
        served += m._ra is not None
        same(xtx, bx[i], (i, labels_kind))
        for a, b in zip(st, bst):
            assert (a is None) == (b is None)
            if a is not None:
                np.testing.assert_allclose(a.double().cpu().numpy(), b[i].double().cpu().numpy(), rtol=1e-11 if dtype is np.float64 else 1e-5)
        xtx.zero_()                                   # the caller owns what it was handed
    assert served >= len(keys) - 1                    # (the read-ahead was on from the first call)
    # the loop again (second pass over the same folds: computed again, not the zeroed slices)
    for i, k in enumerate(keys[:40]):
        v = p.get_validation_indices(k)
        xtx = (m.training_XTX_XTY(v) if M else m.training_XTX(v))[0]
        xtx = xtx[0] if M else xtx
        same(xtx, bx[i], "second pass")
    # a copy of an array and an array changed in place: ordinary route, same numbers to rounding
    v = p.get_validation_indices(keys[41]).copy()
    xtx = m.training_XTX(v)[0]
    assert_normwise(xtx.double(), bx[41].double().cpu().numpy(), 1e-11 if dtype is np.float64 else 1e-5, "copy")
    v = p.get_validation_indices(keys[42])
    m.training_XTX(p.get_validation_indices(keys[41]))          # (read-ahead positioned at fold 42)
    saved = v.copy()
    other = int(p.get_validation_indices(keys[43])[0])
    v[0] = other
    xtx = m.training_XTX(v)[0]
    v[:] = saved
    o = amd.CVMatrix(dtype=dtype, lazy_fit=False)
    o.fit(X, Y, w)
    chk = saved.copy(); chk[0] = other
    assert_normwise(xtx.double(), o.training_XTX(chk.copy())[0].double().cpu().numpy(), 1e-11 if dtype is np.float64 else 1e-5,
                    "changed in place")
    # the reference's raise arrives at the offending call, not before: all weight in one fold
    if labels_kind == "loo":
        w2 = np.zeros(N, dtype=dtype)
        w2[5] = 1.0
        m2 = amd.CVMatrix(dtype=dtype, ddof=0)
        m2.fit(X, Y, w2)
        for i, k in enumerate(keys[:12]):
            v = p.get_validation_indices(k)
            if i == 5:
                with pytest.raises(ValueError, match="non-zero weights"):
                    m2.training_XTX(v)
            else:
                m2.training_XTX(v)


def _oracle_fold(X, Y, w, v, dtype=np.float64):
    from oracle.cvmatrix_oracle import OracleCVMatrix

    o = OracleCVMatrix(dtype=dtype)
    o.fit(X, Y, w)
    return o.training_XTX_XTY(np.array(v, copy=True))


@pytest.mark.serving
@pytest.mark.parametrize("route", ["sweep_loop", "sweep_loop_second_pass", "cached_batch", "read_ahead"])
@pytest.mark.parametrize("change", ["one_interior_index", "sum_preserving_pair"])
def test_indices_changed_in_place_are_never_served_stale(amd, route, change):
    """cvmatrix.py:924-941 gathers from whatever the index array holds at the time of the call.  The
    loop-serving short cuts (one sweep for all folds of a Partitioner, the uploaded batch of an
    earlier pass, the read-ahead over many folds) compare the caller's array EXACTLY with the
    private copy their results were computed from: ONE interior index changed in place -- at a
    position no sampled witness would look at -- or two indices changed so that size, both ends and
    the sum stay what they were, and the call returns the oracle's matrices of the NEW indices."""
    rng = np.random.default_rng(77)
    if route == "read_ahead":
        N, K, M, P = 4000, 40, 3, 40           # > 16 folds of 100 rows: the read-ahead serves the loop
    else:
        N, K, M, P = 6000, 64, 3, 6            # few large folds: one sweep serves the loop
    X, Y, w = rng.random((N, K)), rng.random((N, M)), rng.random(N) + 0.01
    p = amd.Partitioner(np.arange(N) % P)
    keys = list(p.folds_dict)
    m = amd.CVMatrix()
    m.fit(X, Y, w)
    for k in keys[:2]:
        m.training_XTX_XTY(p.get_validation_indices(k))       # the loop is being served
    if route == "read_ahead":
        assert m._ra is not None
    else:
        assert m._sweep is not None and m._sweep_ids is not None
    if route in ("sweep_loop_second_pass", "cached_batch"):
        for k in keys[2:]:
            m.training_XTX_XTY(p.get_validation_indices(k))
        m.fit(X, Y, w)                                       # a second pass: the uploaded batch is kept
        assert p in m._pbatches
    v = p.get_validation_indices(keys[2])
    saved = v.copy()
    n = v.size
    j = n // 2 + 1                                           # (not a multiple of n // 61, not an end)
    assert j % max(1, n // 61) != 0 or n < 122
    if change == "one_interior_index":
        v[j] = int(p.get_validation_indices(keys[3])[7])     # a row of another fold
    else:
        # two entries replaced by rows of the neighbouring folds: size, both ends and the sum unchanged
        v[j], v[j + 2] = int(v[j]) + 1, int(v[j + 2]) - 1
        assert int(v.sum()) == int(saved.sum()) and v[0] == saved[0] and v[-1] == saved[-1]
    try:
        if route == "cached_batch":
            (bx, by), bst = m.training_XTX_XTY_batched(p)    # every fold of the Partitioner, as it is NOW
            xtx, xty, st = bx[2], by[2], tuple(s[2] for s in bst)
        else:
            (xtx, xty), st = m.training_XTX_XTY(v)
        (rx, ry), rst = _oracle_fold(X, Y, w, v)
        assert_normwise(xtx, rx, 1e-10, f"{route}/{change} XTX")
        assert_normwise(xty, ry, 1e-10, f"{route}/{change} XTY")
        assert_stats(st, rst, 1e-10, f"{route}/{change}")
        # and it is NOT the old fold's result
        (ox, _), _ = _oracle_fold(X, Y, w, saved)
        assert np.abs(to_np(xtx) - ox).max() > 1e-8 * np.abs(ox).max()
    finally:
        v[:] = saved
    # restored: served again, the old result
    (xtx, _), _ = m.training_XTX_XTY(v)
    (ox, _), _ = _oracle_fold(X, Y, w, saved)
    assert_normwise(xtx, ox, 1e-10, "restored")


@pytest.mark.serving
def test_reused_output_buffers_never_leak_into_the_served_loop(amd):
    """``reuse_outputs=True`` writes a call's results into buffers keyed by shape; the per-fold loop is served
    from slices the object keeps.  A batched call of the same shape made BETWEEN two calls of the loop must not
    change what the loop's later calls return (the kept slices are the object's own copies)."""
    rng = np.random.default_rng(77)
    N, K, M, P = 6000, 40, 3, 6
    X, Y, w = rng.random((N, K)), rng.random((N, M)), rng.random(N) + 0.05
    p = amd.Partitioner(np.arange(N) % P)
    other = amd.Partitioner((np.arange(N) // 7) % P)       # another partition into P folds: same output shapes
    o = OracleCVMatrix()
    o.fit(X, Y, w)
    for lazy in (True, False):
        m = amd.CVMatrix(reuse_outputs=True, lazy_fit=lazy)
        m.fit(X, Y, w)
        keys = list(p.folds_dict)
        (ax, ay), ast = m.training_XTX_XTY(p.get_validation_indices(keys[0]))
        ax, ay = ax.clone(), ay.clone()
        m.training_XTX_XTY_batched(other)                   # same (P, K, M): the arena's buffers are rewritten
        for k in keys[1:]:
            v = p.get_validation_indices(k)
            (bx, by), bst = m.training_XTX_XTY(v)
            (rx, ry), rst = o.training_XTX_XTY(v)
            assert_normwise(bx, rx, 1e-10, f"fold {k} XTX")
            assert_normwise(by, ry, 1e-10, f"fold {k} XTY")
            assert_stats(bst, rst, what=f"fold {k}")
        (rx, ry), _ = o.training_XTX_XTY(p.get_validation_indices(keys[0]))
        assert_normwise(ax, rx, 1e-10, "fold 0 XTX")


@pytest.mark.serving
def test_serve_loops_off_recomputes_every_call(amd, monkeypatch):
    """``CVMatrix(serve_loops=False)`` / CVM_SERVE_LOOPS=0: no sweep for the loop, no read-ahead, no
    kept batches, no weights identity cache -- every call launches its own kernels on what it is
    handed; same results to rounding."""
    rng = np.random.default_rng(5)
    N, K, M, P = 3000, 48, 2, 5
    X, Y, w = rng.random((N, K)), rng.random((N, M)), rng.random(N) + 0.01
    p = amd.Partitioner(np.arange(N) % P)
    monkeypatch.setenv("CVM_SERVE_LOOPS", "0")
    assert amd.CVMatrix().serve_loops is False
    monkeypatch.delenv("CVM_SERVE_LOOPS")
    on, off = amd.CVMatrix(), amd.CVMatrix(serve_loops=False)
    assert on.serve_loops is True
    on.fit(X, Y, w); off.fit(X, Y, w)
    for k in p.folds_dict:
        v = p.get_validation_indices(k)
        (a, b), sa = on.training_XTX_XTY(v)
        (c, d), sc = off.training_XTX_XTY(v)
        assert off._sweep is None and off._ra is None and off._sweep_ids is None and len(off._pbatches) == 0
        assert_normwise(c, to_np(a), 1e-11, "serve_loops=False XTX")
        assert_normwise(d, to_np(b), 1e-11, "serve_loops=False XTY")
        (rx, ry), rst = _oracle_fold(X, Y, w, v)
        assert_normwise(c, rx, 1e-10, "vs oracle")
        assert_stats(sc, rst, 1e-10, "vs oracle")
    off.training_XTX_XTY_batched(p)
    assert len(off._pbatches) == 0
    q = amd.Partitioner(np.arange(N) % 600)                  # many small folds: no read-ahead either
    for k in list(q.folds_dict)[:20]:
        off.training_XTX(q.get_validation_indices(k))
        assert off._ra is None


def test_partitioner_folds_dict_reassigned_after_construction(amd):
    """A Partitioner whose ``folds_dict`` entries were replaced after construction is read as it is
    (the index matrix it was built with is not consulted)."""
    rng = np.random.default_rng(8)
    N, K, P = 1200, 24, 4
    X, w = rng.random((N, K)), rng.random(N) + 0.01
    p = amd.Partitioner(np.arange(N) % P)
    new0 = np.arange(0, 200)
    p.folds_dict[0] = new0
    m = amd.CVMatrix()
    m.fit(X, None, w)
    bx, _ = m.training_XTX_batched(p)
    from oracle.cvmatrix_oracle import OracleCVMatrix

    o = OracleCVMatrix()
    o.fit(X, None, w)
    assert_normwise(bx[0], o.training_XTX(new0)[0], 1e-10, "reassigned fold")
    assert_normwise(bx[1], o.training_XTX(p.folds_dict[1])[0], 1e-10, "untouched fold")


def test_fold_batch_carried_to_a_model_with_other_weights_is_recounted(amd):
    """A prepared FoldBatch keeps the non-zero-weight counts of the weights it was prepared for; used
    on ANOTHER model (other weights, same number of fits) the counts are made again, so the
    reference's raises (cvmatrix.py:612-630) are decided on the right ones."""
    rng = np.random.default_rng(9)
    N, K = 400, 6
    X = rng.random((N, K))
    w1 = np.ones(N)
    w2 = np.zeros(N); w2[:40] = 1.0                          # all weight inside fold 0
    folds = [np.arange(0, 40), np.arange(40, 400)]
    a, b = amd.CVMatrix(lazy_fit=False), amd.CVMatrix(lazy_fit=False)
    a.fit(X, None, w1); b.fit(X, None, w2)
    batch = a.prepare_folds(folds)
    a.training_XTX_batched(batch)
    with pytest.raises(ValueError, match="greater than zero"):
        b.training_XTX_batched(batch)


def test_two_threads_two_models_two_streams(amd, hip_device):
    """Two threads, each with its own model, Partitioner and HIP stream, running the reference's
    loop at the same time: the process-wide registries that recognise a Partitioner's arrays are
    locked, the library's per-stream work queues keep the launches apart, results equal the
    single-threaded ones bit for bit."""
    import threading

    import torch

    rng = np.random.default_rng(21)
    N, K, M = 5000, 64, 2
    data = [(rng.random((N, K)), rng.random((N, M)), rng.random(N) + 0.01) for _ in range(2)]
    Ps = [5, 8]

    def run(i, stream, out):
        X, Y, w = data[i]
        res = []
        with torch.cuda.stream(stream) if stream is not None else torch.cuda.device(hip_device):
            for rep in range(6):
                p = amd.Partitioner(np.arange(N) % Ps[i])
                m = amd.CVMatrix()
                m.fit(X, Y, w)
                res = [m.training_XTX_XTY(p.get_validation_indices(k))[0] for k in p.folds_dict]
            torch.cuda.current_stream().synchronize()
        out[i] = [(a.clone(), b.clone()) for a, b in res]

    want = [None, None]
    for i in range(2):
        run(i, None, want)
    got = [None, None]
    errs = []

    def guarded(i, st):
        try:
            run(i, st, got)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=guarded, args=(i, torch.cuda.Stream(device=hip_device))) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(2):
        assert len(got[i]) == len(want[i])
        for (a, b), (c, d) in zip(got[i], want[i]):
            assert torch.equal(a, c) and torch.equal(b, d)


def test_more_streams_than_queue_blocks(amd, hip_device):
    """The library owns one piece of device state: 1024 work-queue blocks per device for the persistent Gram
    kernel, one per stream that launches it.  A service that keeps creating streams must not run out of them:
    blocks of streams with nothing in flight change hands (csrc/host.hpp: acquire_queue).  1300 streams here --
    400 of them destroyed at once, 900 kept alive -- each runs a fit; then the first kept streams run again on
    whatever block they are handed now.  Every result equals the first, bit for bit."""
    import torch

    rng = np.random.default_rng(33)
    N, K, M = 1500, 128, 2
    X = torch.as_tensor(rng.random((N, K)), device=hip_device)
    Y = torch.as_tensor(rng.random((N, M)), device=hip_device)
    w = torch.as_tensor(rng.random(N) + 0.01, device=hip_device)
    m = amd.CVMatrix(copy=False, lazy_fit=False)
    m.fit(X, Y, w)
    want = m.XTX.clone()
    torch.cuda.synchronize()
    kept = []
    for i in range(1300):
        st = torch.cuda.Stream(device=hip_device)
        with torch.cuda.stream(st):
            mi = amd.CVMatrix(copy=False, lazy_fit=False)
            mi.fit(X, Y, w)
            got = mi.XTX
        if i % 100 == 0 or i >= 1290:
            st.synchronize()
            assert torch.equal(got, want), i
        if i >= 400:
            kept.append(st)
        else:
            st.synchronize()
            del st
    torch.cuda.synchronize()
    for st in kept[:40]:
        with torch.cuda.stream(st):
            mi = amd.CVMatrix(copy=False, lazy_fit=False)
            mi.fit(X, Y, w)
            got = mi.XTX
        st.synchronize()
        assert torch.equal(got, want)


@pytest.mark.parametrize("dtype", [np.float16, np.longdouble])
def test_reference_dtype_surface_float16_and_float128(amd, dtype):
    """The reference runs float16, float32, float64 and float128 (tests/test_cvmatrix.py:1147-1205).
    float16: inputs rounded to float16 like cvmatrix.py:1146, arithmetic in float32, float16 results
    -- at least as close to the float64 truth as the reference's own float16 arithmetic (the
    oracle run in float16 is the yardstick).  np.longdouble: arithmetic in float64, NumPy results of
    the requested type, 1e-10 norm-wise against the oracle run in long double."""
    import torch

    rng = np.random.default_rng(31)
    N, K, M, P = 240, 20, 3, 4
    X, Y = rng.random((N, K)) + 0.5, rng.random((N, M))
    w = rng.random(N) + 0.1
    folds = [np.arange(f, N, P) for f in range(P)]
    m = amd.CVMatrix(dtype=dtype)
    m.fit(X, Y, w)
    (bx, by), bst = m.training_XTX_XTY_batched(folds)
    (lx, ly), lst = m.training_XTX_XTY(folds[1])
    if dtype is np.float16:
        assert bx.dtype == torch.float16 and lx.dtype == torch.float16 and m.XTX.dtype == torch.float16
        X16, Y16, w16 = (a.astype(np.float16) for a in (X, Y, w))
        truth = OracleCVMatrix(dtype=np.float64)
        truth.fit(X16.astype(np.float64), Y16.astype(np.float64), w16.astype(np.float64))
        ref16 = OracleCVMatrix(dtype=np.float16)
        ref16.fit(X16, Y16, w16)
        eps16 = float(np.finfo(np.float16).eps)
        for f in (1, 3):
            (tx, ty), tst = truth.training_XTX_XTY(folds[f])
            with np.errstate(all="ignore"):
                (rx, ry), _ = ref16.training_XTX_XTY(folds[f])
            for got, t, r in ((bx[f], tx, rx), (by[f], ty, ry)):
                scale = np.abs(t).max()
                err = np.abs(to_np(got).astype(np.float64) - t).max() / scale
                yard = np.nanmax(np.abs(np.asarray(r, dtype=np.float64) - t)) / scale
                assert np.isfinite(err) and err <= 2 * (yard if np.isfinite(yard) else 1.0) + 4 * eps16, (err, yard)
        assert_normwise(lx.float(), to_np(bx[1]).astype(np.float64), 1e-3, "per-fold call")
    else:
        assert isinstance(bx, np.ndarray) and bx.dtype == np.longdouble and isinstance(m.XTX, np.ndarray)
        assert m.XTX.dtype == np.longdouble and lst[0].dtype == np.longdouble
        ref = OracleCVMatrix(dtype=np.longdouble)
        ref.fit(X.astype(np.longdouble), Y.astype(np.longdouble), w.astype(np.longdouble))
        assert_normwise(m.XTX, np.asarray(ref.XTX, dtype=np.float64), 1e-10, "XTX")
        for f in (0, 2):
            (rx, ry), rst = ref.training_XTX_XTY(folds[f])
            assert_normwise(bx[f], np.asarray(rx, dtype=np.float64), 1e-10, "XTX fold")
            assert_normwise(by[f], np.asarray(ry, dtype=np.float64), 1e-10, "XTY fold")
            assert_stats(tuple(s[f] for s in bst), tuple(np.asarray(s, dtype=np.float64) for s in rst), 1e-10, "stats")


def test_backend_numpy_literal_runs_the_reference_call_sequence(amd):
    """``CVMatrix(backend="numpy")`` -- the reference's default literal -- is accepted: ndarray in,
    ndarray out (the arithmetic is the HIP library's)."""
    rng = np.random.default_rng(3)
    X, Y = rng.random((100, 12)), rng.random((100, 2))
    m = amd.CVMatrix(backend="numpy")
    m.fit(X, Y)
    p = amd.Partitioner(np.arange(100) % 5)
    o = OracleCVMatrix()
    o.fit(X, Y)
    for k in p.folds_dict:
        (a, b), st = m.training_XTX_XTY(p.get_validation_indices(k))
        assert isinstance(a, np.ndarray) and isinstance(st[0], np.ndarray)
        (ra, rb), rst = o.training_XTX_XTY(p.get_validation_indices(k))
        assert_normwise(a, ra, 1e-10, "XTX"); assert_normwise(b, rb, 1e-10, "XTY")
    assert isinstance(m.XTX, np.ndarray)
