#!/bin/bash
# round 4: chain_plan_kernel's LEVEL-0 instance from classify_kernel's slots (next read's slot asked for ahead) against gathering by itself; bin_tasks_kernel 32 reads per block
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04f; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== GPU tests =="
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -5 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
for v in "X=1" "MONI_AF_NOPREP=1"; do
  echo "== bench $v =="
  ( export $v; MONI_BENCH_SAVE_INDEX=1 timeout -k 10 600 python bench.py --steps 8 --warmup 2 --no-cpu --no-from-host --no-scaling-base > $OUT/bench_$v.json 2> $OUT/bench_$v.err ) || { tail -5 $OUT/bench_$v.err; exit 1; }
  python - <<PY
import json; d = json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1]); print("$v", d["value"], d["ms_per_step"], d["stages_s_per_step"])
PY
done
echo "== clean per-kernel times =="
bash profiles/clean_times.sh > $OUT/clean_times.txt 2>&1; head -14 $OUT/clean_times.txt
