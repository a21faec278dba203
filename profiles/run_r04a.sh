#!/bin/bash
# round 4, first GPU trip: chain_plan_kernel with four reads per wavefront (16-lane groups) against one read per wavefront.
# bash profiles/run_r04a.sh   (one gpurun call)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04a; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== single-end GPU tests =="
timeout -k 10 900 python -m pytest tests/test_gpu_align.py tests/test_golden.py -m gpu -x -q > $OUT/pytest_se.log 2>&1; rc=$?; tail -5 $OUT/pytest_se.log
[ $rc -ne 0 ] && exit $rc
echo "== bench, groups of 16 (default) =="
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu --no-from-host > $OUT/bench_gw16.json 2> $OUT/bench_gw16.err || exit 1
python - <<PY
import json; d = json.loads(open("$OUT/bench_gw16.json").read().strip().splitlines()[-1]); print("gw16", d["value"], d["ms_per_step"])
PY
echo "== bench, one read per wavefront =="
MONI_AF_GW=64 timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu --no-from-host > $OUT/bench_gw64.json 2> $OUT/bench_gw64.err || exit 1
python - <<PY
import json; d = json.loads(open("$OUT/bench_gw64.json").read().strip().splitlines()[-1]); print("gw64", d["value"], d["ms_per_step"])
PY
echo "== clean per-kernel times, groups of 16 =="
bash profiles/clean_times.sh > $OUT/clean_times_gw16.txt 2>&1; cat $OUT/clean_times_gw16.txt | head -24
