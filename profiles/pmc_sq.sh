#!/bin/bash
# SQ issue counters of one short bench.py run (separate pass, counters only).  bash profiles/pmc_sq.sh <tag>
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu > $OUT/bench_pmc_sq.json 2> $OUT/bench_pmc_sq.log || exit 1
cd $ROOT
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_sq/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"]);
    if (k, key) not in seen: seen.add((k, key)); calls[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
    if "at::native" in k or "rocprim" in k: continue
    n = calls[k]
    print("%-60s calls %3d  " % (k, n) + "  ".join("%s %.3g" % (c.replace("SQ_", ""), x / n) for c, x in sorted(v.items())))
PY
