cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5c/pytest_gpu.txt 2>&1
tail -15 gpurun_out/r5c/pytest_gpu.txt
