cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s
F=gpurun_out/r5s/fuzz_long_final.txt
python -c "
from cvmatrix_amd import _lib
print('library', _lib.load().cvm_version().decode(), '-- one MI355X box: tools/fuzz_all.py 2000 177; tools/fuzz_small.py 2000 178; CVM_MID_INK=1 tools/fuzz_all.py 800 179; CVM_MID_TILE=0 tools/fuzz_all.py 800 180 (mid-size folds through the fused route, fold-major lists); tools/fuzz_pls.py 300 15')" > $F 2>/dev/null
timeout 900 python tools/fuzz_all.py 2000 177 2>&1 | tail -1 >> $F
timeout 900 python tools/fuzz_small.py 2000 178 2>&1 | tail -1 >> $F
CVM_MID_INK=1 timeout 600 python tools/fuzz_all.py 800 179 2>&1 | tail -1 >> $F
CVM_MID_TILE=0 timeout 600 python tools/fuzz_all.py 800 180 2>&1 | tail -1 >> $F
timeout 600 python tools/fuzz_pls.py 300 15 2>&1 | tail -1 >> $F
cat $F
timeout 900 python tools/soak.py > gpurun_out/r5s/soak.txt 2>&1
tail -14 gpurun_out/r5s/soak.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/r5s/smoke.txt; cat gpurun_out/r5s/smoke.txt
