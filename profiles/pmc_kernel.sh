#!/bin/bash
# SQ counters of the kernels matching a regex in one short bench.py run per environment setting (counters only; index cached first).
# bash profiles/pmc_kernel.sh <tag> <kernel regex> "<ENV=.. ENV=..>" ["<ENV..>" ...]
set -o pipefail
TAG=$1; RE=$2; shift 2
CTRS=${MONI_PMC_CTRS:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-from-host --no-scaling-base --inflight 1 > $OUT/bench_build.json 2> $OUT/bench_build.log || exit 1
i=0
for E in "$@"; do
  i=$((i+1))
  echo "== $E =="
  ( export $E; rocprofv3 --pmc $CTRS --kernel-include-regex "$RE" --output-format csv -d $OUT/pmc_$i -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-from-host --no-scaling-base --inflight 1 > $OUT/bench_pmc_$i.json 2> $OUT/bench_pmc_$i.log ) || exit 1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_$i/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); calls[k] += 1
with open("$OUT/pmc_$i.txt", "w") as o:
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
        n = calls[k]
        line = "%-70s x%-3d " % (k, n) + " ".join("%s=%.4g" % (c.replace("SQ_", ""), x / n) for c, x in sorted(v.items()))
        print(line); o.write(line + "\n")
PY
  find $OUT/pmc_$i -name "*.csv" -size +20M -delete
done
