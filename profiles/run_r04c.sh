#!/bin/bash
# round 4: mem_kernel with the 2-bit text (default) against the byte comparison (MONI_MEM_BYTES=1)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04c; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== seeding GPU tests =="
timeout -k 10 900 python -m pytest tests/test_gpu_seed.py tests/test_gpu_align.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -5 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
for v in "X=1" "MONI_MEM_BYTES=1"; do
  echo "== bench $v =="
  ( export $v; MONI_BENCH_SAVE_INDEX=1 timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu --no-from-host > $OUT/bench_$v.json 2> $OUT/bench_$v.err ) || exit 1
  python - <<PY
import json; d = json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1]); print("$v", d["value"], d["ms_per_step"], {k: d[k] for k in d if "seed" in k or "stage" in k})
PY
done
echo "== clean per-kernel times =="
bash profiles/clean_times.sh > $OUT/clean_times.txt 2>&1; head -24 $OUT/clean_times.txt
