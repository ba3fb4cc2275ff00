#!/bin/bash
# Re-collect the rocprofv3 summaries kept under profiles/rN (run on the GPU box through gpurun):
#   bash tools/collect_profiles.sh r1
# One --stats pass over the default headline run, separate --pmc passes over a short one
# (counters serialise kernels), then tools/summarize_rocprof.py condenses them.
set -u
R=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/prof_$R
mkdir -p "$O" "$ROOT/gpurun_out/$R"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --headline-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B > $O/bench_stats.out 2> $O/stats.log
tail -1 $O/bench_stats.out > $ROOT/gpurun_out/$R/bench_stats.json
S="--steps 5 --warmup 2"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $B $S > $O/f.out 2>&1; grep "^{" $O/f.out | tail -1 > $ROOT/gpurun_out/$R/bench_pmc_fetch.json
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $B $S > $O/w.out 2>&1; grep "^{" $O/w.out | tail -1 > $ROOT/gpurun_out/$R/bench_pmc_write.json
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 $B $S > $O/s.out 2>&1; grep "^{" $O/s.out | tail -1 > $ROOT/gpurun_out/$R/bench_pmc_sq.json
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc_tcc -- python3 $B $S > $O/t.out 2>&1; grep "^{" $O/t.out | tail -1 > $ROOT/gpurun_out/$R/bench_pmc_tcc.json
python3 $ROOT/tools/summarize_rocprof.py $O $ROOT/gpurun_out/$R
# the consumer step (PLS) under the same tracer
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pls -- python3 $ROOT/tools/bench_pls.py > $ROOT/gpurun_out/$R/bench_pls.txt 2> $O/pls.log
cp $(ls $O/pls/*/*_kernel_stats.csv | head -1) $ROOT/gpurun_out/$R/pls_kernel_stats.csv 2>/dev/null
cd $ROOT && python3 bench.py > $O/plain.out 2>&1; tail -1 $O/plain.out > $ROOT/gpurun_out/$R/bench_bench_plain.json
tail -3 $O/stats.log
ls -la $ROOT/gpurun_out/$R
