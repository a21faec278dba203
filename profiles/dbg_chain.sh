#!/bin/bash
# where chain_plan_kernel's time goes: the AF_CUTS build (-DAF_CUTS: no stamps, whose atomics distort) cut short after the anchors (4), the chains (8), the lifts (16); one clean
# launch per step (MONI_ALIGN_SUB=1000000), kernel trace.  Results of the cut runs are wrong on purpose.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
for v in ${CUTS:-0 16 8 4}; do
  OUT=$ROOT/gpurun_out/prof_dbgc$v; mkdir -p $OUT
  MONI_ALIGN_SUB=1000000 MONI_AF_DBG=$v MONI_HIP_LIB=$ROOT/moni_align_amd/csrc/libmoni_hip_cuts.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/b.json 2> $OUT/b.log
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
  echo "== MONI_AF_DBG=$v"; python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if "chain_plan" in r["Name"] and "96" in r["Name"]: print("  chain_plan small: calls", r["Calls"], "avg %.3f ms" % (float(r["AverageNs"]) / 1e6))
PY
  find $OUT -name "*kernel_trace.csv" -delete
done
