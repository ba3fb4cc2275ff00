# HBM bytes and L2 hit rate of the mid-size fold kernels at P = 1000 (C3 rows): separate --pmc passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/midpmc; mkdir -p $O
export FOLD_PS=1000
for m in 1 0; do
  export CVM_MID_TILE=$m
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/m$m/stats -- python3 $R/tools/bench_foldsizes.py > $O/m$m.out 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/m$m/pmc_fetch -- python3 $R/tools/bench_foldsizes.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/m$m/pmc_write -- python3 $R/tools/bench_foldsizes.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/m$m/pmc_tcc -- python3 $R/tools/bench_foldsizes.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/m$m/pmc_sq -- python3 $R/tools/bench_foldsizes.py > /dev/null 2>&1
  mkdir -p $R/gpurun_out/midpmc_sum_m$m
  python3 $R/tools/summarize_rocprof.py $O/m$m $R/gpurun_out/midpmc_sum_m$m > /dev/null
  rm -rf $O/m$m
done
