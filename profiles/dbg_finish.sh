#!/bin/bash
# where finish_wave_kernel's time goes: the AF_CUTS build cut short after staging + stitching + lifting (2048), after MD / NM / MAPQ
# (4096), after the segment list (8192); one clean launch per step.  Results of the cut runs are wrong on purpose.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
for v in ${CUTS:-0 8192 4096 2048}; do
  OUT=$ROOT/gpurun_out/prof_dbgf$v; mkdir -p $OUT
  MONI_ALIGN_SUB=1000000 MONI_AF_DBG=$v MONI_HIP_LIB=$ROOT/moni_align_amd/csrc/libmoni_hip_cuts.so timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu > $OUT/b.json 2> $OUT/b.log
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
  echo "== MONI_AF_DBG=$v"; python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if "finish_wave" in r["Name"]: print("  finish_wave: calls", r["Calls"], "avg %.3f ms" % (float(r["AverageNs"]) / 1e6))
PY
  find $OUT -name "*kernel_trace.csv" -delete
done
