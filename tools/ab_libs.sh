#!/bin/bash
# Same-box A/B of builds of the library (rule: never rank builds by timings from different boxes): every build runs
# tools/clock_product.py (>= 2 s of back-to-back steps per workload, the Gram launch by the library's events, the
# in-kernel clock) in turn, twice.   bash tools/ab_libs.sh "C3 C2 C4 C5" default tools/libcvmhip_x.so ...
W=${1:-"C3 C4 C5"}; shift
for i in 1 2; do
  for L in "$@"; do
    if [ "$L" = default ]; then unset CVM_LIB_PATH; else export CVM_LIB_PATH=$L; fi
    python tools/clock_product.py $W --seconds 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); p = j['probe']
    print('%-34s %-6s gram %.4f ms  step %.4f ms  frac %.4f  clock %.0f  cu_time %.4f' % ('$L', j['name'], j['gram_launch_ms'], j['ms_per_step'], j['frac_of_nominal_peak'], p['clock_mhz_median'], p['cu_time_used']))
"
  done
done
