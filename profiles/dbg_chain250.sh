#!/bin/bash
# where chain_plan_kernel's time goes on the configs[4]-shaped workload (250 bp reads, 20 haplotypes: ~200 anchors per read): the AF_CUTS build cut
# short after the anchors (4), the anchor sort (128), the chain DP (256), the chain starts (512), the backtracking (1024), the chains (8), the
# lifts (16); one launch per step (MONI_ALIGN_SUB=1000000), kernel trace.  Results of the cut runs are wrong on purpose.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
ARGS="--base-len 46709983 --haps 20 --read-len 250 --no-cpu --no-from-host"
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2>&1
for v in ${CUTS:-0 16 8 1024 512 256 128 4}; do
  OUT=$ROOT/gpurun_out/prof_c250_$v; mkdir -p $OUT
  MONI_ALIGN_SUB=1000000 MONI_AF_DBG=$v MONI_HIP_LIB=$ROOT/moni_align_amd/csrc/libmoni_hip_cuts.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 2 --warmup 1 $ARGS > $OUT/b.json 2> $OUT/b.log
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
  echo "== MONI_AF_DBG=$v"; python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if "chain_plan" in r["Name"]:
        tag = "small" if "Li96E" in r["Name"] else "big" if "Li256E" in r["Name"] else "huge"
        print("  chain_plan %s: calls %s avg %.3f ms" % (tag, r["Calls"], float(r["AverageNs"]) / 1e6))
PY
  find $OUT -name "*kernel_trace.csv" -delete
done
