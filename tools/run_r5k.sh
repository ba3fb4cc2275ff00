cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
timeout 1500 python -m pytest tests/test_gpu_planner.py -m gpu -q -s > gpurun_out/r5k/planner.txt 2>&1
grep -E "planner|passed|failed" gpurun_out/r5k/planner.txt
