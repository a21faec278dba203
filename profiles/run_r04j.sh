#!/bin/bash
# round 4: extensions' diagonal bound from the 2-bit forms (band_tasks_kernel), plan_kernel with 256-thread blocks; robustness leg (1 % of the reads with an N)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04j; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== quick sanity =="
DBG_MIX=1 timeout -k 10 300 python profiles/dbg/dbg_band.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | head -8
echo "== GPU tests =="
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -5 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && { grep -a -A12 "Error" $OUT/pytest_gpu.log | head -60 | cut -c1-400; exit $rc; }
for v in "X=1" "MONI_AF_DBG=131072"; do
  echo "== bench $v =="
  ( export $v; MONI_AK_PROFILE=1 MONI_BENCH_SAVE_INDEX=1 timeout -k 10 600 python bench.py --steps 8 --warmup 2 --no-cpu --no-from-host --no-scaling-base > $OUT/bench_$v.json 2> $OUT/bench_$v.err ) || { tail -5 $OUT/bench_$v.err; exit 1; }
  python - <<PY
import json; d = json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1]); print("$v", d["value"], d["ms_per_step"], d["stages_s_per_step"])
PY
done
echo "== bench --n-rate 0.01 =="
timeout -k 10 600 python bench.py --steps 8 --warmup 2 --no-cpu --no-from-host --no-scaling-base --n-rate 0.01 > $OUT/bench_nrate.json 2> $OUT/bench_nrate.err || { tail -5 $OUT/bench_nrate.err; exit 1; }
python - <<PY
import json; d = json.loads(open("$OUT/bench_nrate.json").read().strip().splitlines()[-1]); print("n-rate 0.01", d["value"], d["ms_per_step"], d["stages_s_per_step"], d["align"]["reads_taken_by_general_kernel"], d["align"]["handed_over_because"])
PY
echo "== clean per-kernel times =="
bash profiles/clean_times.sh > $OUT/clean_times.txt 2>&1; head -16 $OUT/clean_times.txt
