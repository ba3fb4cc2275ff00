cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5t
for P in 1000 300 100 3000; do
  echo "##### clean build P=$P" >> gpurun_out/r5t/mid128.txt
  timeout 300 tools/mid_probe_clean $P 512 100000 20 2>&1 | grep -E "mid_probe:|as shipped|mid128|differ|first diff" >> gpurun_out/r5t/mid128.txt
done
echo "##### ablation build P=1000" >> gpurun_out/r5t/mid128.txt
timeout 300 tools/mid_probe 1000 512 100000 20 2>&1 | grep -E "mid128" >> gpurun_out/r5t/mid128.txt
cat gpurun_out/r5t/mid128.txt
