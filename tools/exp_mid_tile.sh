# mid-size folds: mid_tile_kernel against the fused route of the Gram kernel (same box, same process order)
#   bash tools/exp_mid_tile.sh   (on the GPU box)
cd "$(dirname "$0")/.."
export FOLD_PS=${FOLD_PS:-100,200,300,500,1000,2000,3000}
for m in 0 1 0 1; do
  echo "== CVM_MID_TILE=$m (CVM_MID_MAXN=${CVM_MID_MAXN:-default})"
  CVM_MID_TILE=$m timeout 300 python tools/bench_foldsizes.py 2>&1 | tail -8
done
