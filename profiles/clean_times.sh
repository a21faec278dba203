#!/bin/bash
# per-kernel times with one launch of every align kernel per step (MONI_ALIGN_SUB=1000000: nothing overlaps), kernel trace;
# median and minimum over the launches of the timed steps (the first third of a kernel's launches - warm-up, first touches - is dropped)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
[ -f /tmp/moni_bench_cache/idx_61420004_12_lifted_0.mfi ] || MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-from-host --no-scaling-base > /dev/null 2>&1
OUT=$ROOT/gpurun_out/prof_clean; rm -rf $OUT; mkdir -p $OUT
MONI_ALIGN_SUB=1000000 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu --no-from-host --no-scaling-base --inflight 1 ${CLEAN_ARGS} > $OUT/b.json 2> $OUT/b.log
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    n = r["Kernel_Name"].replace("void ", ""); n = n[:n.index("(")] if "(" in n else n
    d[n].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows = []
for n, v in d.items():
    v.sort(); x = [t for _, t in v[len(v) // 3:]]
    rows.append((statistics.median(x), min(x), len(v), n))
tot = 0.0
for med, mn, c, n in sorted(rows, reverse=True)[:26]:
    print("%-66s calls %4d  median %8.3f ms  min %8.3f ms" % (n[:66], c, med, mn))
PY
find $OUT -name "*kernel_trace.csv" -delete
