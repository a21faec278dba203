#!/bin/bash
# round 4: kernel trace and SQ counters of the paired bench (with and without -Z), one context
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04z; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-from-host --no-scaling-base --inflight 1 > /dev/null 2>&1
for z in "" "-Z"; do
  tag=paired$z
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- python3 $ROOT/bench.py --paired $z --steps 3 --warmup 1 --no-cpu --no-from-host --inflight 1 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.log || { tail -3 $OUT/bench_$tag.log; exit 1; }
  cp "$(find $OUT/trace_$tag -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_$tag.csv
  python3 - <<PY | tee $OUT/summary_$tag.txt
import csv, json
d = json.loads(open("$OUT/bench_$tag.json").read().strip().splitlines()[-1])
print("== bench.py --paired $z --steps 3 --warmup 1 --inflight 1: %.2f M pairs/s, %.1f ms per step (seeding %.1f, paired kernels + copies %.1f) ==" % (d["value"] / 1e6, d["ms_per_step"], d["stages_s_per_step"]["seed"] * 1e3, d["stages_s_per_step"]["kernel_and_copies"] * 1e3))
rows = list(csv.DictReader(open("$OUT/kernel_stats_$tag.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:24]:
    n = r["Name"].replace("void ", ""); n = n[:n.index("(")] if "(" in n else n
    print("%-76s calls %5s total %9.2f ms avg %9.3f ms" % (n[:76], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
  find $OUT -name "*kernel_trace.csv" -delete
done
echo "== SQ counters, paired =="
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "pe_plan|pe_select|pe_finish|pe_lines|pe_orphan|pe_align" --output-format csv -d $OUT/pmc -- python3 $ROOT/bench.py --paired --steps 1 --warmup 0 --no-cpu --no-from-host --inflight 1 > $OUT/bench_pmc.json 2> $OUT/bench_pmc.log || exit 1
python3 - <<PY | tee $OUT/pmc_sq_paired.txt
import csv, glob, collections
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void ", "")[:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); calls[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n = calls[k]
    print("%-70s x%-3d " % (k, n) + " ".join("%s=%.4g" % (c.replace("SQ_", ""), x / n) for c, x in sorted(v.items())))
PY
rm -rf $OUT/pmc $OUT/trace_*
