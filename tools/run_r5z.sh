cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5z
o=gpurun_out/r5z/prologue.txt
for P in 1000 3000 300; do
  timeout 300 tools/mid_probe_p2 $P 512 100000 10 >> $o 2>&1
done
cat $o
