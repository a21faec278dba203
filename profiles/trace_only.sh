#!/bin/bash
# rocprofv3 kernel trace + stats of a short bench.py run (per-kernel times).  Usage (through gpurun, from the repo root):
#   bash profiles/trace_only.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r02}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu "$@" > $OUT/bench_trace.json 2> $OUT/bench_trace.log || exit 1
cd $ROOT
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -size +20M -delete
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print("%-70s calls %6s  total %9.2f ms  avg %9.3f ms  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
