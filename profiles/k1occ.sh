#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
for v in 4 5 6 8; do
  OUT=$ROOT/gpurun_out/prof_k1occ$v; mkdir -p $OUT
  MONI_ALIGN_SUB=1000000 MONI_AF_K1OCC=$v rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/b.json 2> $OUT/b.log
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
  echo "== K1OCC=$v"; python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'chain_plan' in r['Name']: print('  ', r['Name'][:60], r['Calls'], '%.3f ms' % (float(r['AverageNs'])/1e6))"
  find $OUT -name "*kernel_trace.csv" -delete
done
