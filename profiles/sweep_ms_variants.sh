#!/bin/bash
# ms_lf_kernel tuning sweep on the GPU box: (chains per lane, min waves/SIMD) variants selected by MONI_MS_VARIANT.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sweep_${1:-r01}
mkdir -p $OUT
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --full-path-reads 0 > $OUT/build.json 2> $OUT/build.log || exit 1
for v in 1 2 3 4 5; do
  MONI_MS_VARIANT=$v python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --full-path-reads 0 > $OUT/variant_$v.json 2> $OUT/variant_$v.log
  python3 -c "
import json;d=json.load(open('$OUT/variant_$v.json'));print('variant $v: ms_lf %.3f ms, value %.1f M reads/s, frac %.3f' % (d['kernels_ms']['ms_lf'], d['value']/1e6, d['roofline']['frac']))" >> $OUT/summary.txt
done
cat $OUT/summary.txt
