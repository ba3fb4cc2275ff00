cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
for P in 1000 300 3000; do
  timeout 300 tools/mid_probe $P 512 100000 20 >> gpurun_out/r5d/mid_probe.txt 2>&1
done
cat gpurun_out/r5d/mid_probe.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 -L 2>/dev/null > $GRAFT_REPO_ROOT/gpurun_out/r5d/counters.txt
grep -i -E "dram|mall|TCC_EA0|TCC_REQ|TCC_HIT|TCC_MISS|TCP_TCC" $GRAFT_REPO_ROOT/gpurun_out/r5d/counters.txt | cut -c1-160 | head -60
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_planner.py -m gpu -q -s -k "flag_wait or holds_the_compute or more_streams or planners_plan or behind_torchs or reused_output or validated_again" > gpurun_out/r5d/pytest_sel.txt 2>&1
grep -E "planner|passed|failed|FAILED|Error" gpurun_out/r5d/pytest_sel.txt | tail -30
