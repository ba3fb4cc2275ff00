"""pytest configuration: the `gpu` marker, repo-root imports, fixture helpers."""

import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

STAT_NAMES = ("muX", "sdX", "muY", "sdY")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "serving: looks INSIDE the loop-serving short cuts (nothing to see under "
                                       "CVM_SERVE_LOOPS=0: tools/route_matrix.sh deselects exactly these)")
    config.addinivalue_line("markers", "planner_plan: asserts the planner's own row-split plan (moot under CVM_FORCE_SPLITS)")


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def golden_stats(z, key):
    """Rebuild the 4-tuple (muX, sdX, muY, sdY) with None where the reference gave None."""
    mask = z[f"{key}/mask"]
    return tuple(z[f"{key}/{n}"] if m else None for n, m in zip(STAT_NAMES, mask))


def to_np(a):
    """numpy view of an oracle (ndarray) or product (torch tensor) result."""
    if a is None:
        return None
    if hasattr(a, "detach"):
        return a.detach().cpu().numpy()
    return np.asarray(a)


def assert_normwise(got, ref, tol=1e-10, what=""):
    """The fp64 parity gate of BASELINE.md section 4: max|d| <= tol*max|ref| and
    ||d||_F <= tol*||ref||_F (element-wise relative error is meaningless after the
    cancellation in the centring step; SURVEY.md 7 'hard parts' 2)."""
    got = to_np(got).astype(np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} != {ref.shape}"
    if ref.size == 0:
        return
    d = np.abs(got - ref)
    scale = max(np.abs(ref).max(), np.finfo(np.float64).tiny)
    assert d.max() <= tol * scale, f"{what}: max|d|={d.max():.3e} > {tol}*{scale:.3e}"
    fr = np.linalg.norm(ref)
    assert np.linalg.norm(got - ref) <= tol * max(fr, np.finfo(np.float64).tiny), what


def assert_stats(got, ref, rtol=1e-10, what=""):
    """None pattern must match exactly; values element-wise rtol (well conditioned)."""
    assert len(got) == len(ref) == 4
    for n, g, r in zip(STAT_NAMES, got, ref):
        assert (g is None) == (r is None), f"{what}: None pattern differs at {n}"
        if r is not None:
            g = to_np(g)
            assert g.shape == np.asarray(r).shape, f"{what}:{n} shape {g.shape}"
            np.testing.assert_allclose(g, r, rtol=rtol, atol=0, err_msg=f"{what}:{n}")


@pytest.fixture(scope="session")
def hip_device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
