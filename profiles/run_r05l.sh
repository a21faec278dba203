#!/bin/bash
# round 4: dp_band_kernel with two positions per LDS byte (two blocks per SIMD)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05l; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== sanity =="
DBG_MIX=1 DBG_N_RATE=0.2 timeout -k 10 300 python profiles/dbg/dbg_band.py 2>&1 | grep -v amdgpu.ids | cut -c1-300 > $OUT/sanity.txt || { tail -5 $OUT/sanity.txt; exit 1; }
head -4 $OUT/sanity.txt
if grep -q " [1-9][0-9]* differing" $OUT/sanity.txt; then echo "SANITY FAILED"; exit 1; fi
run() { local name=$1; shift
  timeout -k 10 600 python bench.py --steps 10 --warmup 2 --no-cpu --no-from-host --no-scaling-base --no-single-context "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { tail -5 $OUT/bench_$name.err; return 1; }
  python - <<PY
import json; d = json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"]), round(d["ms_per_step"], 2), {k: round(v * 1e3, 2) for k, v in d["stages_s_per_step"].items()}, d["align"]["roofline"]["cell_slots_run"])
PY
}
echo "== bench =="
run inflight1 --inflight 1 && run inflight2  || exit 1
echo "== clean per-kernel times =="
bash profiles/clean_times.sh > $OUT/clean_times.txt 2>&1; grep -E "dp_|band_tasks" $OUT/clean_times.txt
echo "== GPU tests (align) =="
timeout -k 10 1100 python -m pytest tests/test_gpu_align.py tests/test_gpu_fullsize.py tests/test_golden.py tests/test_gpu_pe.py -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -3 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && { grep -a -B5 -A25 "Error" $OUT/pytest_gpu.log | head -80 | cut -c1-500; exit $rc; }
exit 0
