"""The float32 accuracy contract of the library in one place (imported by tests/test_gpu_parity.py, tools/fuzz_all.py, bench.py).

BASELINE.md section 4: a float32 result may be off from the float64 reference by at most TWICE the error the
reference's own algorithm makes in NumPy float32 on the same inputs (the "yardstick").  Result and yardstick
are both float32 arrays, so the comparison needs an allowance in float32 roundings of the scale on top:

* TWO roundings under the planner's own row-split plans (calibrated in round 4 over the suite and 3 600
  randomised cases: the Gram kernels sum float32 in chains of at most 1024 rows whatever the plan);
* ONE MORE per partial a FORCED plan (``CVM_FORCE_SPLITS=s_off,s_diag``: tests, tools/route_matrix.sh) adds
  to a tile: the finalize kernels add a tile's row-split partials one after the other in float32, each
  addition a rounding of the running sum, and a forced plan makes sums the planner's plans -- which the two
  roundings are calibrated to -- never make (plan 3,5 at the K = 516 three-fold case: error 2.14e-6 against
  2.02e-6 with two roundings, profiles/r4/route_matrix.txt).
"""
import os

import numpy as np

FP32_EPS = float(np.finfo(np.float32).eps)


def forced_plan():
    """(s_off, s_diag) of CVM_FORCE_SPLITS, or None."""
    e = os.environ.get("CVM_FORCE_SPLITS")
    if not e:
        return None
    try:
        so, sd = (int(v) for v in e.split(","))
    except ValueError:
        return None
    return (so, sd) if so >= 1 and sd >= 1 else None


def fp32_roundings() -> int:
    """Roundings of the scale allowed on top of twice the yardstick (see the module docstring)."""
    plan = forced_plan()
    return 2 + (max(plan) - 1 if plan else 0)


def fp32_floor() -> float:
    return fp32_roundings() * FP32_EPS


def fp32_bound(yardstick: float) -> float:
    """Largest norm-wise error a float32 result may have, given the reference algorithm's own float32 error."""
    return 2.0 * yardstick + fp32_floor()
