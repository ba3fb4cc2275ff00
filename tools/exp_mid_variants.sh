# mid_tile_kernel against the fused route; variants are builds under tools/ (CVM_LIB_PATH)
cd $GRAFT_REPO_ROOT
export FOLD_PS=${FOLD_PS:-300,500,1000,3000}
run() { echo "== $*"; env "$@" timeout 200 python tools/bench_foldsizes.py 2>&1 | grep "P="; }
run CVM_MID_TILE=0
run CVM_MID_TILE=1 CVM_MID_MAXN=400
for v in $MID_VARIANTS; do run CVM_MID_TILE=1 CVM_MID_MAXN=400 CVM_LIB_PATH=$GRAFT_REPO_ROOT/tools/libcvmhip_$v.so; done
run CVM_MID_TILE=1 CVM_MID_MAXN=400
