# experiment: wave priorities of the Gram kernel's roles (tools/var/lib_*.so), C3 / C4 / C5 / C2 headline
R=$GRAFT_REPO_ROOT
for W in ${WLS:-C3}; do
for v in base $VARS; do
  if [ $v = base ]; then unset CVM_LIB_PATH; else export CVM_LIB_PATH=$R/tools/var/lib_$v.so; fi
  S="--steps 200 --warmup 30"; [ $W != C3 ] && [ $W != C2 ] && S="--steps 5 --warmup 2"
  python3 $R/bench.py --headline-only --workload $W $S 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W $v', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
done
