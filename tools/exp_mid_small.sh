# where mid_tile_kernel takes over from the direct small-fold kernels (below) and hands over to the fused Gram
# route (above): folds of 8 .. 256 contiguous rows, three shapes;  bash tools/exp_mid_small.sh [f64|f32]
cd $GRAFT_REPO_ROOT
DT=${1:-f64}
run() { echo "== $*"; env "$@" 2>&1 | grep "n_val\|CVM_SMALL"; }
for shape in "512 16" "1024 8" "2048 4"; do
  export NVS=${NVS:-8,16,24,32,33,48,64,100,128,160,200,256}
  run CVM_MID_TILE=0 timeout 300 python tools/exp_small_limit.py $shape $DT
  run CVM_MID_TILE=1 CVM_MID_MINN=1 CVM_MID_MAXN=256 timeout 300 python tools/exp_small_limit.py $shape $DT
done
