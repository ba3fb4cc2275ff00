cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5u
for V in 0 1 2; do
for P in 1000 300; do
  echo "##### finish prio $V P=$P" >> gpurun_out/r5u/mid128.txt
  timeout 300 tools/mid_probe_p$V $P 512 100000 20 2>&1 | grep -E "as shipped \(again|mid128 dbg|differ" | head -3 >> gpurun_out/r5u/mid128.txt
done; done
cat gpurun_out/r5u/mid128.txt
