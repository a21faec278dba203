#!/bin/bash
# VALU / SALU / LDS instruction counts of chain_plan_kernel and finish_wave_kernel phase by phase: the -DAF_CUTS build cut short at
# MONI_AF_DBG = <bit> (results of the cut runs are wrong on purpose), one counter pass each.  bash profiles/pmc_cuts.sh <tag> <cut> <cut> ...
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-from-host --no-scaling-base --inflight 1 > $OUT/bench_build.json 2> $OUT/bench_build.log || exit 1
for v in "$@"; do
  MONI_AF_DBG=$v MONI_HIP_LIB=$ROOT/moni_align_amd/csrc/libmoni_hip_cuts.so timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
      --kernel-include-regex "chain_plan|finish_wave" --output-format csv -d $OUT/pmc_$v -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-from-host --no-scaling-base --inflight 1 > $OUT/b_$v.json 2> $OUT/b_$v.log || { echo "cut $v failed"; tail -3 $OUT/b_$v.log; continue; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_$v/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    if "96, 48" not in k and "finish_wave" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); calls[k] += 1
with open("$OUT/cuts.txt", "a") as o:
    for k, c in sorted(agg.items()):
        n = calls[k]
        line = "cut %-6s %-40s x%-2d " % ("$v", k[5:45], n) + " ".join("%s=%.4g" % (x.replace("SQ_", ""), y / n) for x, y in sorted(c.items()))
        print(line); o.write(line + "\n")
PY
  rm -rf $OUT/pmc_$v
done
