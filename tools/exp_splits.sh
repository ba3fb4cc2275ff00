mkdir -p gpurun_out/r2b
python -m pytest tests -m gpu -x -q > gpurun_out/r2b/pytest.log 2>&1; tail -5 gpurun_out/r2b/pytest.log
for sp in "" "5,5" "4,7" "4,6" "4,8" "3,5" "8,14" "7,11" "5,8"; do
  CVM_FORCE_SPLITS=$sp python bench.py --headline-only --steps 100 --warmup 20 > gpurun_out/r2b/h_$sp.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2b/h_$sp.json").read().strip().splitlines()[-1])
print("splits=[$sp]", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["parity"][:60])
PY
done
