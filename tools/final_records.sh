#!/bin/bash
# The round's closing records with the FINAL library (run through gpurun): the GPU suite, smoke, the randomised
# sweeps, the race screens, and the profile set.   bash tools/final_records.sh r6
R=${1:-r6}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
D=gpurun_out/$R; mkdir -p $D
V=$(python3 -c "from cvmatrix_amd import _lib; print(_lib.load().cvm_version().decode())" 2>/dev/null)
python3 -m pytest tests -q -m gpu 2>&1 | tail -3 > $D/pytest_gpu_summary.txt
( python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1; echo "smoke rc $?" ) > $D/smoke.txt
{ echo "library $V -- one MI355X box: tools/fuzz_all.py 2000 277; tools/fuzz_small.py 2000 278; CVM_MID_TILE=0 tools/fuzz_all.py 800 280; CVM_VALIDATE_WEIGHTS=sync tools/fuzz_all.py 800 281; tools/fuzz_pls.py 300 25; CVM_RESIDENT=1 tools/fuzz_small.py 600 283";
  python3 tools/fuzz_all.py 2000 277 2>&1 | tail -1; python3 tools/fuzz_small.py 2000 278 2>&1 | tail -1;
  CVM_MID_TILE=0 python3 tools/fuzz_all.py 800 280 2>&1 | tail -1; CVM_VALIDATE_WEIGHTS=sync python3 tools/fuzz_all.py 800 281 2>&1 | tail -1;
  python3 tools/fuzz_pls.py 300 25 2>&1 | tail -1; CVM_RESIDENT=1 python3 tools/fuzz_small.py 600 283 2>&1 | tail -1; } > $D/fuzz_long_final.txt
{ python3 tools/soak.py 2>&1 | grep -v amdgpu.ids; python3 tools/soak_pls.py 2>&1 | tail -3; } > $D/soak.txt
bash tools/collect_profiles.sh $R > gpurun_out/collect_$R.log 2>&1
tail -2 $D/pytest_gpu_summary.txt; cat $D/smoke.txt | tail -2; cat $D/fuzz_long_final.txt; tail -4 $D/soak.txt
