#!/bin/bash
# round 4: default bench line with two contexts in flight; paired path (with and without -Z) with the new DP kernels; counters of finish_wave_kernel
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04p; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== bench default (full) =="
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
python - <<PY
import json; d = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1]); print("default", round(d["value"]), round(d["ms_per_step"], 2), d["roofline"]["frac"], d["roofline"].get("kernel_alone"), d.get("single_context"), d["cpu_baseline"]["value"], d["cpu_baseline"]["sam_identical_on_sample"])
PY
pe() { local name=$1; shift
  timeout -k 10 900 python bench.py --paired --steps 6 --warmup 1 --no-cpu --no-from-host "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { tail -5 $OUT/bench_$name.err; return 1; }
  python - <<PY
import json; d = json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"]), round(d["ms_per_step"], 2), d["stages_s_per_step"], d["pairs_taken_by_pe_align_kernel"], d["handed_over_because"], d.get("single_context"))
PY
}
echo "== paired =="
pe pe_inflight1 --inflight 1 && pe pe_inflight2 --inflight 2 && pe pe_z_inflight1 --inflight 1 -Z && pe pe_z_inflight2 --inflight 2 -Z || exit 1
echo "== counters: finish_wave_kernel, chain_plan_kernel =="
bash profiles/pmc_kernel.sh r04p "finish_wave|chain_plan_kernel<af_wave_tt<96|plan_kernel|ms_lf" "X=1" 2>&1 | tail -12
