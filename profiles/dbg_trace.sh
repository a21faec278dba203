#!/bin/bash
# timing experiments with the AF_PROFILE build: MONI_AF_DBG variants of dp_lane_kernel (cached index)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
for v in 0 1 2; do
  OUT=$ROOT/gpurun_out/prof_dbg$v; mkdir -p $OUT
  MONI_AF_DBG=$v MONI_HIP_LIB=$ROOT/moni_align_amd/csrc/libmoni_hip_prof.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/b.json 2> $OUT/b.log
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
  echo "== MONI_AF_DBG=$v"; grep "dp_lane\|chain_plan\|finish_kernel" $f | awk -F, '{printf "%s calls %s avg %.3f ms\n", substr($1,1,50), $2, $4/1e6}'
  find $OUT -name "*kernel_trace.csv" -delete
done
