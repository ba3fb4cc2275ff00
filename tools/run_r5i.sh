cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "flag_wait or holds_the_compute or route_forcing or mid or fused or randomised or fuzz" > gpurun_out/r5i/pytest_sel.txt 2>&1
tail -12 gpurun_out/r5i/pytest_sel.txt
for e in "CVM_MID_PREPASS=0" "CVM_MID_PREPASS=1"; do
  echo "== $e" >> gpurun_out/r5i/foldsizes.txt
  env $e FOLD_PS=300,500,1000,2000,3000 timeout 300 python tools/bench_foldsizes.py 2>&1 | grep "P=" >> gpurun_out/r5i/foldsizes.txt
done
for e in "CVM_MID_MAXN=400" "CVM_MID_MAXN=600" "CVM_MID_TILE=0"; do
  echo "== $e" >> gpurun_out/r5i/foldsizes.txt
  env $e FOLD_PS=100,200,300,500 timeout 300 python tools/bench_foldsizes.py 2>&1 | grep "P=" >> gpurun_out/r5i/foldsizes.txt
done
cat gpurun_out/r5i/foldsizes.txt
