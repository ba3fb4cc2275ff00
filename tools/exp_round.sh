# regression round: fold-size sweep + full bench at C3/C4/C5 (+C2)
out=gpurun_out/$1; mkdir -p $out
python tools/bench_foldsizes.py > $out/fold_size_sweep.txt 2>/dev/null; cat $out/fold_size_sweep.txt
for w in C3 C2 C4 C5; do
  extra=""; [ $w = C3 ] || extra="--steps 5 --warmup 2 --no-cpu-baseline"
  [ $w = C2 ] && extra="--steps 50 --warmup 10 --no-cpu-baseline"
  timeout 900 python bench.py --workload $w $extra > $out/bench_$w.json 2> $out/bench_$w.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$w.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("$w", "value", d["value"], "ms/step", d["ms_per_step"], "gram ms", r["avg_launch_ms"], "frac", r["frac"], "| two-stage", d.get("two_stage_ms_per_step"), "fit gram", r["two_stage_fit_gram_avg_launch_ms"], "fold gram", r["two_stage_fold_gram_avg_launch_ms"], "| per-fold-call", d["per_fold_call_folds_per_s"], "|", d["parity"][:30])
    for k,v in (d.get("supplementary_hbm_regime") or {}).items(): print("    ", k, v["ms"], v.get("roofline",{}).get("frac"))
except Exception as e:
    print("$w failed", e); print(open("$out/bench_$w.err").read()[-800:])
PY
done
