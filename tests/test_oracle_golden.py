"""CPU: the oracle (oracle/cvmatrix_oracle.py) against the reference's golden vectors.

This is the pin that lets the GPU parity tests trust the oracle: every vector in
tests/golden/ was produced by the reference itself (tests/golden/make_golden.py)."""

import numpy as np
import pytest

import parity_cases as pc
from conftest import load_json, load_npz
from oracle.cvmatrix_oracle import (
    OracleCVMatrix,
    OraclePartitioner,
    benchmark_inputs,
    complement_indices,
    naive_training_matrices,
)

TOL = 1e-11  # oracle vs reference fast path, norm-wise


def make(flags, ddof):
    return OracleCVMatrix(*flags, ddof=ddof)


def test_g1_inline_fixtures():
    pc.run_g1(make, OraclePartitioner, TOL)


def test_g2_readme_quickstart():
    pc.run_g2(make, OraclePartitioner, TOL)


def test_g3_flag_sweep():
    for name in pc.g3_cases():
        pc.run_g3_case(name, make, OraclePartitioner, TOL)


def test_g3_leave_one_out_sweep():
    """The reference's whole leave-one-out sweep (tests/test_cvmatrix.py:1357-1396): 128 cases."""
    cases = pc.g3loo_cases()
    assert len(cases) == 128
    for name in cases:
        pc.run_g3loo_case(name, make, TOL)


def test_g3_naive_restatement_matches_reference_naive():
    """The oracle's direct (training-index) computation vs the reference's NaiveCVMatrix."""
    z = load_npz("g3_sweep.npz")
    p = OraclePartitioner(z["folds"])
    for name in pc.g3_cases():
        if name.startswith("loo_") or not name.endswith("y1"):
            continue
        fl = tuple(c == "1" for c in name[1:5])
        weighted, ddof = name[7] == "1", int(name[10])
        X = z["Xw"] if weighted else z["X"]
        w = z["w"] if weighted else None
        for fi, f in enumerate(p.folds_dict):
            t = complement_indices(p, f)
            (nx, ny), _ = naive_training_matrices(X, z["Y"], w, t, *fl, ddof)
            k = f"{name}/fold{fi}/naive/joint"
            np.testing.assert_allclose(nx, z[f"{k}/XTX"], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(ny, z[f"{k}/XTY"], rtol=1e-9, atol=1e-9)


def test_g4_example_zero_weight_and_str_label():
    pc.run_g4(make, OraclePartitioner, TOL)


def test_g5_error_messages():
    pc.run_g5(OracleCVMatrix, OraclePartitioner)


def test_g7_none_pattern():
    pc.run_g7(OracleCVMatrix)


@pytest.mark.parametrize("name", ["c3", "c4s"])
def test_g6_digest_full_shapes(name):
    """BASELINE.json shapes: the oracle reproduces the reference's per-fold digests."""
    z = load_npz("g6_digest.npz")
    meta = load_json("g6_digest_meta.json")[name]
    X, Y, w, folds = benchmark_inputs(meta["N"], meta["K"], meta["M"], meta["P"])
    m = OracleCVMatrix(*meta["flags"], ddof=1, copy=False)
    m.fit(X, Y, w if meta["weighted"] else None)
    p = OraclePartitioner(folds)
    for f in z[f"{name}/fold_labels"][:3]:
        (xtx, xty), st = m.training_XTX_XTY(p.get_validation_indices(int(f)))
        pc.check_digest(z, name, int(f), xtx, xty, st, 1e-10)
