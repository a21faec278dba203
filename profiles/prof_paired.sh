#!/bin/bash
# kernel trace of the paired mode of bench.py (staged paired kernels + pe_align_kernel over what they hand over)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_paired; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --paired --pairs ${PAIRS:-400000} --steps 2 --warmup 1 --no-cpu > $OUT/b.json 2> $OUT/b.log
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    n = r["Name"].replace("void ", ""); n = n[:n.index("(")] if "(" in n else n
    print("%-60s calls %5s total %9.2f ms avg %9.3f ms" % (n[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
find $OUT -name "*kernel_trace.csv" -delete
tail -2 $OUT/b.json | cut -c1-600
