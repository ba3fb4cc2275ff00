cd $GRAFT_REPO_ROOT
export FOLD_PS=500,1000,3000
run() { echo "== $*"; env "$@" timeout 200 python tools/bench_foldsizes.py 2>&1 | grep "P="; }
run CVM_MID_TILE=0
run CVM_MID_TILE=0 FOLD_CONTIG=1
run CVM_MID_TILE=1
run CVM_MID_TILE=1 FOLD_CONTIG=1
