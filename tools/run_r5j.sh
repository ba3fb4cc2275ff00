cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "flag_wait or route_forcing" > gpurun_out/r5j/pytest_sel.txt 2>&1
tail -4 gpurun_out/r5j/pytest_sel.txt
for e in "CVM_MID_PREPASS=0" "CVM_MID_PREPASS=1"; do
  echo "== $e" >> gpurun_out/r5j/foldsizes.txt
  env $e FOLD_PS=500,1000,2000,3000 timeout 300 python tools/bench_foldsizes.py 2>&1 | grep "P=" >> gpurun_out/r5j/foldsizes.txt
done
cat gpurun_out/r5j/foldsizes.txt
