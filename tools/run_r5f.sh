cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
for P in 1000 300 100 3000; do
  for V in 0 1; do
    echo "##### CVM_MID_SDMA=$V P=$P" >> gpurun_out/r5f/mid_probe.txt
    timeout 300 tools/mid_probe_sdma$V $P 512 100000 20 2>&1 | grep -E "mid_tile dbg= 0|chmax=1 |differ" | head -4 >> gpurun_out/r5f/mid_probe.txt
  done
done
cat gpurun_out/r5f/mid_probe.txt
