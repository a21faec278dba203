#!/bin/bash
# bench.py with alternative builds of the library: bash profiles/try_libs.sh lib1.so lib2.so ...
MONI_BENCH_SAVE_INDEX=1 python3 bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> gpurun_out/sweep_build.log || exit 1
for l in "$@"; do
  MONI_HIP_LIB=$PWD/moni_align_amd/csrc/$l timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l:', round(d['value']/1e6,2), 'M reads/s', round(d['ms_per_step'],1), 'ms', d['align']['kernels_ms_per_step_summed'])" || exit 1
done
