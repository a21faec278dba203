#!/bin/bash
# round 4: -Z with a LEVEL-2 instance of pe_plan_kernel (pairs with more chains than AF_MAX_CHAINS); the driver's bench command
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04q; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
pe() { local name=$1; shift
  MONI_BENCH_SAVE_INDEX=1 timeout -k 10 900 python bench.py --paired --steps 6 --warmup 1 --no-cpu --no-from-host "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { tail -5 $OUT/bench_$name.err; return 1; }
  python - <<PY
import json; d = json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"]), round(d["ms_per_step"], 2), d["stages_s_per_step"], d["pairs_taken_by_pe_align_kernel"], d["handed_over_because"], d.get("single_context"))
PY
}
echo "== paired -Z =="
pe pe_z_inflight1 --inflight 1 -Z && pe pe_z_inflight2 --inflight 2 -Z || exit 1
echo "== paired tests =="
timeout -k 10 1100 python -m pytest tests/test_gpu_pe.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/pytest_pe.log 2>&1; rc=$?; tail -5 $OUT/pytest_pe.log
[ $rc -ne 0 ] && { grep -a -B5 -A25 "Error" $OUT/pytest_pe.log | head -80 | cut -c1-500; exit $rc; }
echo "== the driver's command =="
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err || { tail -5 $OUT/bench_driver.err; exit 1; }
python - <<PY
import json; d = json.loads(open("$OUT/bench_driver.json").read().strip().splitlines()[-1]); print("driver cmd", round(d["value"]), round(d["ms_per_step"], 2), d["roofline"]["frac"], d["single_context"]["ms_per_step"], d["cpu_baseline"]["value"], d["cpu_baseline"]["sam_identical_on_sample"], d.get("scaling_base", {}).get("value"))
PY
