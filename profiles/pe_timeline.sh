#!/bin/bash
# timeline of one paired step: start / end of every kernel launch of the last step relative to its first kernel (rocprofv3 --kernel-trace)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pe_timeline; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --paired --pairs ${PAIRS:-1000000} --steps 1 --warmup 1 --no-cpu --no-from-host > $OUT/b.json 2> $OUT/b.log
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - <<PY | tee $OUT/timeline.txt
import csv
rows = list(csv.DictReader(open("$f")))
def nm(r):
    n = r["Kernel_Name"].replace("void ", ""); n = n[:n.index("(")] if "(" in n else n
    return n[:44]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r), r.get("Queue_Id", "?")) for r in rows), key=lambda x: x[0])
# the last step: from the last pack_kernel (first kernel of seeding) on
starts = [i for i, e in enumerate(ev) if e[2].startswith("pack_kernel")]
lo = starts[-1]
t0 = ev[lo][0]
for s, e, n, q in ev[lo:]:
    if (e - s) < 150000 and not n.startswith(("pe_", "dp_lane", "ms_lf", "mem_k", "occ_k", "gather")) and not (int("${TL_FROM:-0}") * 1e6 <= s - t0 <= int("${TL_TO:-0}") * 1e6): continue
    print("%8.2f -> %8.2f ms  (%7.2f)  q%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
print("step span %.2f ms" % ((max(e for s, e, n, q in ev[lo:]) - t0) / 1e6))
PY
find $OUT -name "*kernel_trace.csv" -delete
