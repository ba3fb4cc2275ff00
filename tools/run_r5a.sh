cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
export CVM_LIB_PATH=$GRAFT_REPO_ROOT/tools/libcvmhip_base.so
FOLD_PS=30,100,200,300,500,1000,3000 timeout 300 python tools/bench_foldsizes.py > gpurun_out/r5a/foldsizes_base.txt 2>&1
for P in 100 300 1000; do
  echo "== P=$P" >> gpurun_out/r5a/fused_stamps_base.txt
  CVM_MID_TILE=0 timeout 200 python tools/fused_stamps.py tools/libcvmhip_stamps.so $P >> gpurun_out/r5a/fused_stamps_base.txt 2>&1
done
cat gpurun_out/r5a/foldsizes_base.txt gpurun_out/r5a/fused_stamps_base.txt
