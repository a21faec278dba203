#!/bin/bash
# per-kernel times with one launch of every align kernel per step (MONI_ALIGN_SUB=1000000: nothing overlaps), kernel trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
OUT=$ROOT/gpurun_out/prof_clean; mkdir -p $OUT
MONI_ALIGN_SUB=1000000 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/b.json 2> $OUT/b.log
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows = sorted(csv.DictReader(open("$f")), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    n = r["Name"].replace("void ", ""); n = n[:n.index("(")] if "(" in n else n
    print("%-62s calls %4s avg %8.3f ms" % (n[:62], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
find $OUT -name "*kernel_trace.csv" -delete
