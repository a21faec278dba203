#!/bin/bash
# HBM fetch / L2 read requests of ms_lf_kernel alone (counters only, one pass each), default bench workload
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_mslf; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
for c in FETCH_SIZE TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-include-regex "ms_lf|mem_kernel" --output-format csv -d $OUT/$c -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > $OUT/$c.json 2> $OUT/$c.log
  f=$(find $OUT/$c -name "*counter_collection.csv" | head -1)
  python3 - <<PY
import csv, collections
agg = collections.defaultdict(list)
try:
    for r in csv.DictReader(open("$f")):
        agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print("$c", k, "launches", len(v), "mean %.4g" % (sum(v) / len(v)))
except Exception as e: print("$c failed", e)
PY
done
