cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5v
run() { echo "##### P=$1 K=$2 N=$3" >> gpurun_out/r5v/mid128.txt; timeout 300 tools/mid_probe_p0 $1 $2 $3 10 2>&1 | grep -E "mid_probe:|as shipped \(again|mid128 dbg|differ" | head -4 >> gpurun_out/r5v/mid128.txt; }
run 6000 512 96000
run 3000 512 100000
run 2000 512 100000
run 1500 512 100000
run 1250 512 100000
run 1000 512 100000
run 700 512 100000
run 500 512 100000
run 900 1024 30000
run 300 1024 30000
run 200 2048 10000
cat gpurun_out/r5v/mid128.txt
