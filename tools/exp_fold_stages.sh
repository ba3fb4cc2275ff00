R=$GRAFT_REPO_ROOT
for rep in 1 2; do for lib in default tools/libcvmhip_notwo.so tools/libcvmhip_fs32.so tools/libcvmhip_fs64.so; do
  if [ $lib = default ]; then unset CVM_LIB_PATH; else export CVM_LIB_PATH=$R/$lib; fi
  python bench.py --workload C5 --steps 5 --warmup 2 --brief --no-live-traffic --device-data 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C5 $lib', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['parity'][:120])"
done; done
unset CVM_LIB_PATH
