# experiment: statistics-only fold stage, units per fold forced (CVM_COL_SPLITS)
mkdir -p gpurun_out/cs; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${VARS:-0}; do
  if [ $v = 0 ]; then unset CVM_COL_SPLITS; else export CVM_COL_SPLITS=$v; fi
  echo "== CVM_COL_SPLITS=$v"
  timeout 300 python3 $R/tools/bench_stats.py $SHAPES 2>&1 | grep "N="
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cs/$v -o p -- python3 $R/tools/bench_stats.py $SHAPES > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
for f in glob.glob("$R/gpurun_out/cs/$v/**/p_kernel_trace.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    d=collections.defaultdict(list)
    for r in rows:
        n=r["Kernel_Name"]
        if "colstats" in n or "fold_stats" in n:
            d[n[:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for n,v in d.items():
        for i in range(0,len(v),21):
            seg=v[i:i+21]
            print("  %-60s n=%d med %.1f us min %.1f"%(n,len(seg),sorted(seg)[len(seg)//2],min(seg)))
PY
done
