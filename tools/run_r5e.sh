cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5e
for P in 1000 300; do
  echo "##### ablation build" >> gpurun_out/r5e/mid_probe.txt
  timeout 300 tools/mid_probe $P 512 100000 20 2>&1 | grep -v "^mid_tile dbg= *[1-9]" >> gpurun_out/r5e/mid_probe.txt
  echo "##### clean build" >> gpurun_out/r5e/mid_probe.txt
  timeout 300 tools/mid_probe_clean $P 512 100000 20 2>&1 | grep -E "as shipped|as built|against" >> gpurun_out/r5e/mid_probe.txt
done
cat gpurun_out/r5e/mid_probe.txt
