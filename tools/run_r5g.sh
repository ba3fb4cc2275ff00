cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
for P in 1000 300; do
  timeout 300 tools/mid_probe_sdma0 $P 512 100000 20 2>&1 | grep -E "mid_probe:|mid_tile " >> gpurun_out/r5g/occupancy.txt
done
cat gpurun_out/r5g/occupancy.txt
