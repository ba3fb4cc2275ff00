R=$GRAFT_REPO_ROOT
for lib in default tools/libcvmhip_sr4.so tools/libcvmhip_sr2.so tools/libcvmhip_sr16.so; do for f in 8 4 2; do
  if [ $lib = default ]; then unset CVM_LIB_PATH; else export CVM_LIB_PATH=$R/$lib; fi
  echo "== $lib fpr=$f"; CVM_SMALL_FPR=$f timeout 200 python tools/bench_hbm.py quick 2>&1 | grep "LOOCV"
done; done
