# experiments: pin the row-split plan (CVM_FORCE_SPLITS="s_off,s_diag") and time the C3 headline
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2d/pytest.log 2>&1; tail -5 gpurun_out/r2d/pytest.log
for sp in "" "5,5" "4,7" "4,6" "4,8" "3,5" "8,14" "8,13" "7,11" "5,8" "3,6" "6,10"; do
  CVM_FORCE_SPLITS=$sp timeout 300 python bench.py --headline-only --steps 100 --warmup 20 > gpurun_out/r2d/h_$sp.json 2>gpurun_out/r2d/h_$sp.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r2d/h_$sp.json").read().strip().splitlines()[-1])
    print("splits=[$sp]", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["parity"][:40])
except Exception as e:
    print("splits=[$sp] failed", e)
PY
done
