#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
for v in ${GRIDS:-24 12 6 48 96}; do
  OUT=$ROOT/gpurun_out/prof_fg$v; mkdir -p $OUT
  MONI_ALIGN_SUB=1000000 MONI_AF_FINGRID=$v timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu > $OUT/b.json 2> $OUT/b.log
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
  echo "== waves per CU $v"; python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if "finish_wave" in r["Name"]: print("  finish_wave: avg %.3f ms" % (float(r["AverageNs"]) / 1e6))
PY
  find $OUT -name "*kernel_trace.csv" -delete
done
