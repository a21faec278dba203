#!/bin/bash
# bench.py under environment settings: bash profiles/sweep_env.sh "A=1" "A=2 B=3" ...
MONI_BENCH_SAVE_INDEX=1 python3 bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> gpurun_out/sweep_build.log || exit 1
for s in "$@"; do
  env $s timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$s:', round(d['value']/1e6,2), 'M reads/s', round(d['ms_per_step'],1), 'ms', d['align']['kernels_ms_per_step_summed'])" || exit 1
done
