cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5y
timeout 300 tools/xcd_stack_probe > gpurun_out/r5y/xcd_stack.txt 2>&1
cat gpurun_out/r5y/xcd_stack.txt
