cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
O=gpurun_out/r5n/route_matrix_rest.txt
for e in "CVM_FUSED_PREPASS=1" "CVM_FUSED_ORDER=1" "CVM_MID_INK=1"; do
  echo "== $e" >> $O
  env $e timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -q \
      -k "not bench_command and not plan and not full_size_properties and not forced_split and not randomised" 2>&1 | grep -E "^FAILED|passed|failed" | tail -8 >> $O
done
for e in "CVM_NO_FUSED=1" "CVM_FORCE_FALLBACK=1" "CVM_MID_OWNSTATS=1"; do
  echo "== $e (the two tests that failed in the first call, fixed)" >> $O
  env $e timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "flag_wait" 2>&1 | grep -E "^FAILED|passed|failed" | tail -4 >> $O
done
echo "== CVM_SERVE_LOOPS=0 (the test that failed in the first call, fixed)" >> $O
CVM_SERVE_LOOPS=0 timeout 600 python -m pytest tests/test_gpu_boundary.py -m "gpu" -q -k "behind_torchs or validated_again" 2>&1 | grep -E "^FAILED|passed|failed" | tail -4 >> $O
cat $O
echo "##### stamps C3" > gpurun_out/r5n/stamps.txt
timeout 300 python tools/stamps.py tools/libcvmhip_stamps.so >> gpurun_out/r5n/stamps.txt 2>&1
echo "##### stamps C5 scaled (N=50000 K=4096 M=1 P=5 f32)" >> gpurun_out/r5n/stamps.txt
STAMP_N=50000 STAMP_K=4096 STAMP_M=1 STAMP_P=5 STAMP_DTYPE=f32 timeout 300 python tools/stamps.py tools/libcvmhip_stamps.so >> gpurun_out/r5n/stamps.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r5n/stamps.txt
