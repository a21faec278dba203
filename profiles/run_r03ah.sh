set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03ah
timeout -k 10 600 python3 -m pytest tests/test_gpu_pe.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -2
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
MONI_AK_PROFILE=1 timeout -k 10 500 python3 bench.py --paired --steps 5 --warmup 1 --no-cpu > gpurun_out/r03ah/bench_paired.json 2> gpurun_out/r03ah/bench_paired.log
grep "paired batch" gpurun_out/r03ah/bench_paired.log | tail -2
python3 - <<PY
import json
d = json.loads(open("gpurun_out/r03ah/bench_paired.json").read().strip().splitlines()[-1])
print(round(d["value"] / 1e6, 3), "M pairs/s", round(d["ms_per_step"], 1), "ms", d.get("from_host"))
PY
