cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5w
o=gpurun_out/r5w/order.txt
for v in "0 0" "2 0" "2 4" "0 4" "2 2"; do set -- $v
  echo "##### CVM_SMALL_NOREMAP=$1 CVM_SMALL_FPB=$2" >> $o
  if [ "$2" = "0" ]; then CVM_SMALL_NOREMAP=$1 timeout 600 python tools/bench_hbm.py >> $o 2>&1
  else CVM_SMALL_NOREMAP=$1 CVM_SMALL_FPB=$2 timeout 600 python tools/bench_hbm.py >> $o 2>&1; fi
done
cat $o
