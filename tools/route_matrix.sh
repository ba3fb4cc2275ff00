#!/bin/bash
# The GPU parity suite under every route-forcing switch of the library (read once per process):
# forced row-split plans in both orders, the two-stage route instead of the fused epilogue, the
# general Gram kernel instead of the LDS-DMA one, the separate finalize kernels instead of the
# one-call sweep's, the tile kernel instead of the whole-rows kernel for tiny folds.
#   bash tools/route_matrix.sh        (on the GPU box; about two minutes per switch)
cd "$(dirname "$0")/.."
for e in "CVM_FORCE_SPLITS=3,5" "CVM_FORCE_SPLITS=7,2" "CVM_NO_FUSED=1" "CVM_FORCE_FALLBACK=1" \
         "CVM_NO_SWEEP_MERGE=1" "CVM_NO_DIRECT=1" "CVM_PAD=0"; do
  echo "== $e"
  env $e timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -q \
      -k "not bench_command and not plan and not full_size_properties and not forced_split and not randomised" 2>&1 | tail -2
done
