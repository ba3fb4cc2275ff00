cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "flag_wait or holds_the_compute or route_forcing or forced_split or mid or fused" > gpurun_out/r5b/pytest_sel.txt 2>&1
tail -15 gpurun_out/r5b/pytest_sel.txt
for o in 2 1; do
  echo "== CVM_FUSED_ORDER=$o" >> gpurun_out/r5b/foldsizes.txt
  CVM_FUSED_ORDER=$o FOLD_PS=100,200,300,500 timeout 300 python tools/bench_foldsizes.py 2>&1 | grep "P=" >> gpurun_out/r5b/foldsizes.txt
  echo "== CVM_FUSED_ORDER=$o CVM_MID_TILE=0" >> gpurun_out/r5b/foldsizes.txt
  CVM_MID_TILE=0 CVM_FUSED_ORDER=$o FOLD_PS=500,1000,3000 timeout 300 python tools/bench_foldsizes.py 2>&1 | grep "P=" >> gpurun_out/r5b/foldsizes.txt
done
cat gpurun_out/r5b/foldsizes.txt
