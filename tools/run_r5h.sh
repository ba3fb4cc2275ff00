cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $O/pmc_rd -- python3 $R/tools/bench_hbm.py quick > $O/rd.out 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum --output-format csv -d $O/pmc_wr -- python3 $R/tools/bench_hbm.py quick > $O/wr.out 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUBBLE_sum --output-format csv -d $O/pmc_hit -- python3 $R/tools/bench_hbm.py quick > $O/hit.out 2>&1
python3 - <<'PY'
import csv, glob, os, collections, json
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/r5h"
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for cc in glob.glob(O+"/pmc_*/*/*_counter_collection.csv"):
    per=collections.defaultdict(float)
    for row in csv.DictReader(open(cc)):
        per[(row["Dispatch_Id"], row["Kernel_Name"][:90], row["Counter_Name"])]+=float(row["Counter_Value"])
    for (_,k,c),v in per.items(): acc[k][c].append(v)
out={}
for k,cs in acc.items():
    if "small_" not in k: continue
    out[k]={c:{"mean":sum(v)/len(v),"n":len(v)} for c,v in cs.items()}
json.dump(out, open(O+"/hbm_regime_dram_counters.json","w"), indent=1)
for k,d in out.items():
    print(k)
    for c,v in sorted(d.items()): print("   %-28s %14.0f  (n=%d)"%(c,v["mean"],v["n"]))
PY
tail -5 $O/rd.out
