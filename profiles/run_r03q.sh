# round 3: full GPU test suite, default bench, paired bench (CPU baseline included) and its kernel trace on the build with the paired path's lines
# written on the GPU, 262144-pair chunks, two-level pe_plan_kernel, 16 hardware queues
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03q
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03q/pytest_gpu.log 2>&1 || { tail -60 gpurun_out/r03q/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r03q/pytest_gpu.log
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 500 python3 bench.py --steps 5 --warmup 2 > gpurun_out/r03q/bench_default.json 2> gpurun_out/r03q/bench_default.log
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03q/bench_default.json").read().strip().splitlines()[-1])
print("default", round(d["value"] / 1e6, 2), "M reads/s", round(d["ms_per_step"], 1), "ms", {k: round(v, 2) for k, v in d["kernels_ms"].items() if k != "note"}, round(d["roofline"]["frac"], 3), d.get("from_host", {}).get("value"), d.get("from_host", {}).get("two_contexts", {}).get("value"))
PY
timeout -k 10 500 python3 bench.py --paired --steps 4 --warmup 1 > gpurun_out/r03q/bench_paired.json 2> gpurun_out/r03q/bench_paired.log || { tail -20 gpurun_out/r03q/bench_paired.log; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03q/bench_paired.json").read().strip().splitlines()[-1])
print("paired", round(d["value"] / 1e6, 3), "M pairs/s", round(d["ms_per_step"], 1), "ms", d["stages_s_per_step"], d.get("cpu_baseline"))
PY
PAIRS=1000000 bash profiles/prof_paired.sh 2>&1 | tail -22
