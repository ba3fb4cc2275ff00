cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5z
o=gpurun_out/r5z/early.txt
for P in 1000 300 3000 2000 6000; do
  for b in mid_probe_clean mid_probe_p1 mid_probe_p0; do
    echo "##### $b P=$P" >> $o
    timeout 300 tools/$b $P 512 100000 10 2>&1 | grep -E "mid_probe:|as shipped|differ" | head -4 >> $o
  done
done
echo "##### stamps (ablate build, G early)" >> $o
timeout 300 tools/mid_probe_p2 1000 512 100000 10 2>&1 | grep -E "mid_probe:|as shipped|stamps" | head -4 >> $o
cat $o
