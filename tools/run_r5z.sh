cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5z
o=gpurun_out/r5z/routes.txt
export FOLD_PS=100,125,150,200,250,300,400,500,700,1000
echo "##### default" >> $o; timeout 600 python tools/bench_foldsizes.py 2>&1 | grep "P=" >> $o
echo "##### CVM_MID_MAXN=1100" >> $o; CVM_MID_MAXN=1100 timeout 600 python tools/bench_foldsizes.py 2>&1 | grep "P=" >> $o
echo "##### CVM_MID_MAXN=1100 CVM_MID_INK=1" >> $o; CVM_MID_MAXN=1100 CVM_MID_INK=1 timeout 600 python tools/bench_foldsizes.py 2>&1 | grep "P=" >> $o
cat $o
