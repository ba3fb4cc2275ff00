#!/bin/bash
# The GPU parity suite under every route-forcing switch of the library (read once per process):
# forced row-split plans in both orders, the two-stage route instead of the fused epilogue, the
# general Gram kernel instead of the LDS-DMA one, the separate finalize kernels instead of the
# one-call sweep's, the tile kernel instead of the whole-rows kernel for tiny folds, no compaction /
# inline statistics, no loop serving, the direct kernels for folds of up to 64 / 128 rows, the mid-size tile
# kernel off / from one row per fold / up to 1000 rows per fold, the round-6 resident route for float32 folds of <= 16 rows.
#   bash tools/route_matrix.sh        (on the GPU box; about two minutes per switch)
cd "$(dirname "$0")/.."
for e in "CVM_FORCE_SPLITS=3,5" "CVM_FORCE_SPLITS=7,2" "CVM_NO_FUSED=1" "CVM_FORCE_FALLBACK=1" \
         "CVM_NO_SWEEP_MERGE=1" "CVM_NO_DIRECT=1" "CVM_PAD=0" "CVM_NO_COMPACT=1" "CVM_NO_INLINE_STATS=1" \
         "CVM_SERVE_LOOPS=0" "CVM_SMALL_MAXN=128" "CVM_SMALL_MAXN=64" "CVM_MID_TILE=0" "CVM_MID_MINN=1" "CVM_MID_MAXN=1000" \
         "CVM_FUSED_PREPASS=1" "CVM_FUSED_ORDER=1" "CVM_VALIDATE_WEIGHTS=sync" "CVM_RESIDENT=1" "CVM_RESIDENT=0"; do
  echo "== $e"
  mark="gpu"
  extra=""
  # (with loop serving off, the tests that look INSIDE the serving have nothing to see: they carry the
  #  `serving` marker.  The float32 cases run under the forced split plans too, no case deselected: the gate's
  #  allowance is derived from the plan in effect -- one float32 rounding per partial a forced plan adds to a
  #  tile's ordered sum, cvmatrix_amd/fp32_gate.py.)
  case "$e" in
    CVM_SERVE_LOOPS=0) mark="gpu and not serving" ;;
  esac
  env $e timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m "$mark" -q \
      -k "not bench_command and not plan and not full_size_properties and not forced_split and not randomised$extra" 2>&1 | grep -E "^FAILED|passed|failed" | tail -8
done
# the planner against its forced neighbours at the 5 % bar (the suite itself only fails at 15 %: a wall-clock
# assertion does not belong in `pytest -m gpu -x`)
echo "== planner neighbours (CVM_PLANNER_STRICT=1)"
CVM_PLANNER_STRICT=1 timeout 900 python -m pytest tests/test_gpu_planner.py -m gpu -q -s 2>&1 | grep -E "planner|passed|failed" | tail -12
