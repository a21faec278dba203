#!/bin/bash
# round 4: full GPU suite on the build with the 2-bit mem_kernel (pointer prefetch), finer DP bins and cut extension targets; bench with the per-bin
# counts; instruction counts of chain_plan_kernel / finish_wave_kernel phase by phase (AF_CUTS build)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04d; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== GPU tests =="
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -5 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
echo "== bench =="
MONI_AK_PROFILE=1 MONI_BENCH_SAVE_INDEX=1 timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu --no-from-host > $OUT/bench.json 2> $OUT/bench.err || exit 1
python - <<PY
import json; d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("stages_s_per_step"))
PY
grep -a "DP problems per bin" $OUT/bench.err | tail -4
grep -a "staged kernels, sub-batch" $OUT/bench.err | tail -4
echo "== clean per-kernel times =="
bash profiles/clean_times.sh > $OUT/clean_times.txt 2>&1; head -24 $OUT/clean_times.txt
echo "== instruction counts by phase =="
bash profiles/pmc_cuts.sh r04d_cuts 0 4 128 256 512 1024 8 16 16384 2048 4096 8192 2>&1 | grep "^cut"
