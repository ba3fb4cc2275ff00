cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "float32 or f32 or fp32 or randomised or fuzz or full_size or c5 or C5" > gpurun_out/r5q/pytest_f32.txt 2>&1
tail -5 gpurun_out/r5q/pytest_f32.txt
for i in 1 2; do
timeout 600 python bench.py --workload C5 --brief --steps 5 --warmup 2 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C5', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('parity'))" >> gpurun_out/r5q/c5.txt
done

import json,sys
timeout 300 python tools/bench_f32_midsize.py >> gpurun_out/r5q/c5.txt 2>&1

cat gpurun_out/r5q/c5.txt
