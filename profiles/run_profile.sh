#!/bin/bash
# Profiling recipe for the GPU box (run through gpurun from the repo root):
#   bash profiles/run_profile.sh <tag>
# 1) builds + caches the full-scale flat index, 2) rocprofv3 kernel trace + stats of bench.py,
# 3) separate PMC passes (FETCH_SIZE / WRITE_SIZE / TCC request counters, SQ issue counters), as MI355X_MICROARCH.md prescribes.
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench_build.json 2> $OUT/bench_build.log || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu > $OUT/bench_trace.json 2> $OUT/bench_trace.log || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench_pmc_fetch.json 2> $OUT/bench_pmc_fetch.log || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench_pmc_write.json 2> $OUT/bench_pmc_write.log || exit 1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench_pmc_tcc.json 2> $OUT/bench_pmc_tcc.log || echo "tcc pass failed" >> $OUT/bench_pmc_tcc.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu > $OUT/bench_pmc_sq.json 2> $OUT/bench_pmc_sq.log || echo "sq pass failed" >> $OUT/bench_pmc_sq.log
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
cd $ROOT
python3 profiles/summarize_profile.py $OUT > $OUT/summary.txt 2>&1 || true
# keep the merged-back directory small: only summaries and per-kernel stats
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
