#!/bin/bash
# round 4: extensions and gap fills banded too (dp_band_kernel), histogram atomics of global_band_kernel aggregated
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04h; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== GPU tests (align, golden, pe, fullsize) =="
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -5 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
for v in "X=1" "MONI_AF_NOPLANK=1" "MONI_AF_DBG=131072" "MONI_AF_DBG=196608"; do
  echo "== bench $v =="
  ( export $v; MONI_AK_PROFILE=1 MONI_BENCH_SAVE_INDEX=1 timeout -k 10 600 python bench.py --steps 8 --warmup 2 --no-cpu --no-from-host --no-scaling-base > $OUT/bench_$v.json 2> $OUT/bench_$v.err ) || { tail -5 $OUT/bench_$v.err; exit 1; }
  python - <<PY
import json; d = json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1]); print("$v", d["value"], d["ms_per_step"], d["stages_s_per_step"]); print(d["align"]["roofline"]["padding"]["useful_over_slots"], d["align"]["roofline"]["cells"], d["align"]["roofline"]["cells_after_cut"], d["align"]["roofline"]["cell_slots_run"])
PY
done
grep -a "DP problems per bin" "$OUT/bench_X=1.err" | tail -1
echo "== clean per-kernel times =="
bash profiles/clean_times.sh > $OUT/clean_times.txt 2>&1; head -22 $OUT/clean_times.txt
