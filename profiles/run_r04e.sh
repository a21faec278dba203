#!/bin/bash
# round 4: band widths of the global problems (histogram), default bench line of the committed build with the new legs, clean per-kernel times (median)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04e; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== bench (default flags, no CPU baseline) =="
MONI_AK_PROFILE=1 MONI_BENCH_SAVE_INDEX=1 timeout -k 10 900 python bench.py --no-cpu > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python - <<PY
import json; d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print(d["value"], d["reads_per_s"], d["ms_per_step"], d.get("stages_s_per_step")); print(d.get("scaling_base")); print(d["align"]["roofline"]); print(d["setup_s"])
PY
grep -a "global problems by band" $OUT/bench.err | tail -4
echo "== clean per-kernel times =="
bash profiles/clean_times.sh > $OUT/clean_times.txt 2>&1; head -26 $OUT/clean_times.txt
