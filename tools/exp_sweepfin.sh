# experiment: variants of sweep_finish_kernel (tools/var/lib_*.so), kernel time under rocprofv3
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in $VARS; do
  export CVM_LIB_PATH=$R/tools/var/lib_$v.so
  rm -rf /tmp/sw_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sw_$v -o p -- python3 $R/bench.py --headline-only --steps 50 --warmup 5 > /dev/null 2>&1
  echo "== $v: $(grep sweep_finish /tmp/sw_$v/p_kernel_stats.csv | cut -d, -f4,6,7 | tail -1)"
done
