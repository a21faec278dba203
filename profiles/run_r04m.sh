#!/bin/bash
# round 4: wildcard problems in queues of their own, scored by a WILDC instance of dp_lane_kernel (the plain instances as before the wildcard planes: no scratch)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04m; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== quick sanity =="
DBG_MIX=1 timeout -k 10 300 python profiles/dbg/dbg_band.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | head -2
echo "== GPU tests =="
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -5 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && { grep -a -B5 -A25 "Error" $OUT/pytest_gpu.log | head -80 | cut -c1-500; exit $rc; }
echo "== bench default =="
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 600 python bench.py --steps 8 --warmup 2 --no-cpu --no-from-host --no-scaling-base > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
python - <<PY
import json; d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1]); print("default", d["value"], d["ms_per_step"], d["stages_s_per_step"])
PY
for nr in 0.01 0.05; do
echo "== bench --n-rate $nr =="
timeout -k 10 600 python bench.py --steps 8 --warmup 2 --no-from-host --no-scaling-base --n-rate $nr --cpu-seconds 6 > $OUT/bench_nrate_$nr.json 2> $OUT/bench_nrate_$nr.err || { tail -5 $OUT/bench_nrate_$nr.err; exit 1; }
python - <<PY
import json; d = json.loads(open("$OUT/bench_nrate_$nr.json").read().strip().splitlines()[-1]); print("n-rate $nr", d["value"], d["ms_per_step"], d["stages_s_per_step"], d["align"]["reads_taken_by_general_kernel"], d["align"]["handed_over_because"], d["cpu_baseline"]["sam_identical_on_sample"])
PY
done
echo "== clean per-kernel times =="
bash profiles/clean_times.sh > $OUT/clean_times.txt 2>&1; head -22 $OUT/clean_times.txt
