cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
bash tools/route_matrix.sh > gpurun_out/r5m/route_matrix.txt 2>&1
cat gpurun_out/r5m/route_matrix.txt
