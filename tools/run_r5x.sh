cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5x
python -c "from cvmatrix_amd import build as b; print('lib', b.source_hash(), b._embedded_hash_without_loading())" > gpurun_out/r5x/hash.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r5x/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5x/smoke.txt 2>&1; echo "smoke rc $?" >> gpurun_out/r5x/smoke.txt
timeout 600 python bench.py > gpurun_out/r5x/bench.json 2> gpurun_out/r5x/bench.err; echo "bench rc $?" >> gpurun_out/r5x/bench.err
cat gpurun_out/r5x/hash.txt gpurun_out/r5x/pytest_gpu.txt; tail -3 gpurun_out/r5x/smoke.txt; tail -2 gpurun_out/r5x/bench.err; head -c 600 gpurun_out/r5x/bench.json
