#!/bin/bash
# round 4: GPU tests with dp_wave_kernel; contexts in flight (seeding of one step beside the align kernels of another)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04o; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
run() { # name, bench args..., env via ENVV
  local name=$1; shift
  env $ENVV MONI_BENCH_SAVE_INDEX=1 timeout -k 10 600 python bench.py --steps 12 --warmup 2 --no-cpu --no-from-host --no-scaling-base "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { tail -5 $OUT/bench_$name.err; return 1; }
  python - <<PY
import json; d = json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"]), round(d["ms_per_step"], 2), {k: round(v * 1e3, 2) for k, v in d["stages_s_per_step"].items()}, "lf ms", round(d["roofline"]["avg_launch_ms"], 2))
PY
}
echo "== bench =="
ENVV="X=1" run inflight1 --inflight 1 && ENVV="X=1" run inflight2 --inflight 2 && ENVV="X=1" run inflight3 --inflight 3 && ENVV="MONI_ALIGN_SUB=333334" run inflight2_sub333k --inflight 2 && ENVV="MONI_ALIGN_SUB=500000" run inflight2_sub500k --inflight 2 || exit 1
echo "== GPU tests =="
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -5 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && { grep -a -B5 -A25 "Error" $OUT/pytest_gpu.log | head -80 | cut -c1-500; exit $rc; }
exit 0
