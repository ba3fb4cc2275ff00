#!/bin/bash
# Re-collect the rocprofv3 summaries kept under profiles/rN (run on the GPU box through gpurun):
#   bash tools/collect_profiles.sh r2
# One --stats pass over the default headline run, separate --pmc passes over a short one
# (counters serialise kernels), then tools/summarize_rocprof.py condenses them.
set -u
R=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/prof_$R
D=$ROOT/gpurun_out/$R
mkdir -p "$O" "$D"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --headline-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B > $O/bench_stats.out 2> $O/stats.log
grep "^{" $O/bench_stats.out | tail -1 > $D/bench_stats.json
S="--steps 5 --warmup 2"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $B $S > $O/f.out 2>&1; grep "^{" $O/f.out | tail -1 > $D/bench_pmc_fetch.json
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $B $S > $O/w.out 2>&1; grep "^{" $O/w.out | tail -1 > $D/bench_pmc_write.json
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 $B $S > $O/s.out 2>&1; grep "^{" $O/s.out | tail -1 > $D/bench_pmc_sq.json
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc_tcc -- python3 $B $S > $O/t.out 2>&1; grep "^{" $O/t.out | tail -1 > $D/bench_pmc_tcc.json
python3 $ROOT/tools/summarize_rocprof.py $O $D > $D/pmc_derived.json
# the other BASELINE workloads: --stats only
for W in C2 C4 C5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$W -- python3 $B --workload $W $S > $O/st_$W.out 2> $O/st_$W.log
  grep "^{" $O/st_$W.out | tail -1 > $D/bench_stats_$W.json
  cp $(ls $O/stats_$W/*/*_kernel_stats.csv | head -1) $D/kernel_stats_$W.csv 2>/dev/null
done
# the HBM-bound regime (small folds): --stats and FETCH / WRITE passes over tools/bench_hbm.py (back-to-back timing, as bench.py)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/small -- python3 $ROOT/tools/bench_hbm.py quick > $D/bench_hbm.txt 2> $O/small.log
cp $(ls $O/small/*/*_kernel_stats.csv | head -1) $D/small_folds_kernel_stats.csv 2>/dev/null
mkdir -p $O/smallpmc
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/smallpmc/pmc_fetch -- python3 $ROOT/tools/bench_hbm.py quick > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/smallpmc/pmc_write -- python3 $ROOT/tools/bench_hbm.py quick > /dev/null 2>&1
mkdir -p $D/small_tmp; python3 $ROOT/tools/summarize_rocprof.py $O/smallpmc $D/small_tmp > /dev/null; mv $D/small_tmp/pmc_summary.json $D/small_folds_pmc_summary.json; rm -rf $D/small_tmp
python3 $ROOT/tools/bench_hbm.py > $D/bench_hbm_full.txt 2>/dev/null
# the statistics-only call and the float32 mid-size folds
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_only -- python3 $ROOT/tools/bench_stats.py > $D/bench_statistics.txt 2> $O/stats_only.log
cp $(ls $O/stats_only/*/*_kernel_stats.csv | head -1) $D/statistics_kernel_stats.csv 2>/dev/null
python3 $ROOT/tools/bench_f32_midsize.py > $D/bench_f32_midsize.txt 2>/dev/null
# the consumer step (PLS) under the same tracer
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pls -- python3 $ROOT/tools/bench_pls.py > $D/bench_pls.txt 2> $O/pls.log
cp $(ls $O/pls/*/*_kernel_stats.csv | head -1) $D/pls_kernel_stats.csv 2>/dev/null
cd $ROOT
python3 bench.py > $O/plain.out 2>&1; grep "^{" $O/plain.out | tail -1 > $D/bench_bench_plain.json
python3 tools/bench_foldsizes.py > $D/fold_size_sweep.txt 2>/dev/null
# mid-size folds (P = 1000) under the counters: mid_tile_kernel behind the statistics pre-pass
FOLD_PS=1000 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/mid_fetch -- python3 $ROOT/tools/bench_foldsizes.py > /dev/null 2>&1
FOLD_PS=1000 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/mid_write -- python3 $ROOT/tools/bench_foldsizes.py > /dev/null 2>&1
# (round 5: LDS bank conflicts against all LDS cycles, matrix-core busy cycles, for P = 1000 -- mid_tile_kernel -- and P = 300 -- the fused route)
FOLD_PS=1000,300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mid_sq -- python3 $ROOT/tools/bench_foldsizes.py > /dev/null 2>&1
mkdir -p $O/midpmc; mv $O/mid_fetch $O/midpmc/pmc_fetch; mv $O/mid_write $O/midpmc/pmc_write; mv $O/mid_sq $O/midpmc/pmc_sq
mkdir -p $D/mid_tmp; python3 $ROOT/tools/summarize_rocprof.py $O/midpmc $D/mid_tmp > /dev/null; mv $D/mid_tmp/pmc_summary.json $D/midsize_P1000_pmc_summary.json; rm -rf $D/mid_tmp
python3 tools/emulate_scaling.py --workloads C3 --out $D/emulated_scaling_C3 > $O/emu_C3.log 2>&1; python3 tools/emulate_scaling.py --workloads C4 --comm-us 0,60 --out $D/emulated_scaling_C4 > $O/emu_C4.log 2>&1
rm -f $D/benchmark_protocol_hip.csv; python3 tools/benchmark_protocol.py --csv $D/benchmark_protocol_hip.csv > $D/benchmark_protocol.log 2>&1
python3 tools/power_probe.py C3 C3fit C3fold C3two C4 C4fit C5 2>/dev/null > $D/power_probe.txt
# round 6: the clock under the PRODUCT kernel (cvm_clock_probe) with rocm-smi's sclk beside it, the calibration of the
# stamps on a bare MFMA loop, and the float32 loop probe (MFMA shapes, wave-block shapes, what a vector instruction costs)
python3 tools/clock_product.py C3 C3fit C3fold C2 C4 C5 --seconds 3 2>/dev/null > $D/clock_product.txt
./tools/mfma_peak 2000000 > $D/mfma_peak_clock_calibration.txt 2>&1
./tools/f32_loop_probe 3000 > $D/f32_loop_probe.txt 2>&1
for p in "500:250" "240:280,280:120" "240:280,240:135" "480:145,560:65"; do ./tools/dispatch_probe "$p" > /tmp/dp.txt; python3 tools/dispatch_analyze.py /tmp/dp.txt | sed -n 1,7p | cut -c1-400; echo; done > $D/dispatch_probe.txt 2>&1
tail -3 $O/stats.log
ls -la $D
