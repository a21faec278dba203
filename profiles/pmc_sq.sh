#!/bin/bash
# SQ issue counters of the align-stage kernels in one short bench.py run (separate pass, counters only; the index is built once
# and cached so that the profiled run only loads it).  bash profiles/pmc_sq.sh <tag> [counters...]
set -o pipefail
TAG=${1:-r02}; shift
CTRS=${@:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
echo "building + caching the index"; MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > $OUT/bench_build.json 2> $OUT/bench_build.log || exit 1
echo "counter pass: $CTRS"
rocprofv3 --pmc $CTRS --kernel-include-regex "chain_plan|dp_lane|select_kernel|traceback|finish_kernel|global_task|align_kernel|ms_lf|mem_kernel|occ_kernel" --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > $OUT/bench_pmc_sq.json 2> $OUT/bench_pmc_sq.log || exit 1
cd $ROOT
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_sq/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:44]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); calls[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    n = calls[k]
    print("%-44s x%-3d " % (k, n) + " ".join("%s=%.3g" % (c.replace("SQ_", ""), x / n) for c, x in sorted(v.items())))
PY
