"""GPU: the row-split planner against its neighbours.

The planner's constants (csrc/host.hpp: per-stage costs, the diagonal tiles' relative cost, what extra partials
cost the finalize kernels) were fitted to 23 measured plans at ONE shape, C3.  This test measures, at the C3 shape
and at scaled C4 / C5 shapes, the sweep's Gram launch + finalize under the planner's own (s_off, s_diag) and under
forced neighbours of it (``cvm_debug_force_splits``: one atomic word in the library, no environment reads per call)
and reports the table.

A wall-clock assertion inside ``pytest -m gpu -x`` is one noisy neighbour away from hiding every test after it, so
the suite only fails when the planner's plan is more than 15 % slower than the best neighbour (a planner that has
gone wrong, not a noisy box); the 5 % bar the planner is held to is asserted under ``CVM_PLANNER_STRICT=1``
(tools/route_matrix.sh runs that; profiles/r*/planner_neighbours.txt is its record)."""

import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [
    # name, N, K, M, P, dtype
    ("C3", 100000, 512, 16, 10, np.float64),
    ("C4 scaled (N / 4, 16 folds)", 250000, 1024, 32, 16, np.float64),
    ("C5 scaled (N / 4, 5 folds)", 50000, 4096, 1, 5, np.float32),
    # the slowest rank's share of C3 at 8 / 4 / 2 GPUs (2 / 3 / 5 of the ten folds): the multi-GPU step's Gram launch
    ("C3r8 (2 folds: a rank of 8)", 20000, 512, 16, 2, np.float64),
    ("C3r4 (3 folds: a rank of 4)", 30000, 512, 16, 3, np.float64),
    ("C3r2 (5 folds: a rank of 2)", 50000, 512, 16, 5, np.float64),
]


def _step_ms(amd, torch, X, Y, w, labels, dtype, reps=12):
    """Median time of fit + batched call over the partition (one sweep: Gram launch + finalize) under the
    plan in effect, by events on the stream."""
    m = amd.CVMatrix(dtype=dtype, copy=False, lazy_fit=True, reuse_outputs=True, trust_tensor_versions=True)
    p = amd.Partitioner(labels)
    m.fit(X, Y, w)
    b = m.prepare_folds(p)
    for _ in range(3):
        m.fit(X, Y, w)
        m.training_XTX_XTY_batched(b)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        m.fit(X, Y, w)
        m.training_XTX_XTY_batched(b)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


@pytest.mark.planner_plan
@pytest.mark.parametrize("name,N,K,M,P,dtype", SHAPES, ids=[s[0].split()[0] for s in SHAPES])
def test_the_planners_plan_is_not_beaten_by_its_neighbours(name, N, K, M, P, dtype):
    import torch

    import cvmatrix_amd as amd
    from cvmatrix_amd import _lib

    if os.environ.get("CVM_FORCE_SPLITS"):
        pytest.skip("a plan is forced from outside")
    lib = _lib.load()
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype is np.float64 else torch.float32
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    X = torch.rand((N, K), dtype=tdt, device=dev, generator=g)
    Y = torch.rand((N, M), dtype=tdt, device=dev, generator=g)
    w = torch.rand((N,), dtype=tdt, device=dev, generator=g) + 0.01
    labels = np.arange(N) % P
    info = (C.c_int64 * 8)()
    cdt = _lib.CVM_F64 if dtype is np.float64 else _lib.CVM_F32
    assert lib.cvm_plan_fold(P, (N + P - 1) // P, K, M, cdt, 0x3F, 1 << 40, info) == 0
    so, sd = int(info[0]), int(info[6])
    cand = [(so, sd)]
    for dso, dsd in ((-1, 0), (1, 0), (0, -1), (0, 1), (1, 1), (-1, -1), (2, 2), (-2, -2), (3, 3), (2, 0), (0, 2)):
        c = (so + dso, sd + dsd)
        if c[0] >= 1 and c[1] >= 1 and c not in cand:
            cand.append(c)
    times = {}
    try:
        # warm the device up on the planner's plan, then two interleaved rounds over all plans
        _step_ms(amd, torch, X, Y, w, labels, dtype, reps=20)
        for rnd in range(2):
            for c in cand:
                assert lib.cvm_debug_force_splits(*((0, 0) if c == (so, sd) else c)) == 0
                t = _step_ms(amd, torch, X, Y, w, labels, dtype)
                times[c] = min(times.get(c, 1e9), t)
    finally:
        lib.cvm_debug_force_splits(0, 0)
    best = min(times, key=times.get)
    report = ", ".join(f"{c[0]}/{c[1]}: {t:.4f}" for c, t in sorted(times.items(), key=lambda kv: kv[1]))
    print(f"{name}: planner {so}/{sd} {times[(so, sd)]:.4f} ms; all (ms): {report}")
    bar = 1.05 if os.environ.get("CVM_PLANNER_STRICT", "0") != "0" else 1.15
    assert times[(so, sd)] <= bar * times[best], (
        f"{name}: the planner's plan {so}/{sd} takes {times[(so, sd)]:.4f} ms, plan {best[0]}/{best[1]} "
        f"{times[best]:.4f} ms ({report})")
