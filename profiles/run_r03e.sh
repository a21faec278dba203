# round 3: regrouped fast rows (a step reads 16 bytes of its row, a threshold jump 16 more) + the paired mode of bench.py
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
timeout -k 10 600 python3 -m pytest tests/test_gpu_seed.py tests/test_golden.py -m gpu -x -q > gpurun_out/r03e/pytest_subset.log 2>&1 || { tail -40 gpurun_out/r03e/pytest_subset.log; exit 1; }
tail -2 gpurun_out/r03e/pytest_subset.log
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/r03e/bench_default.json 2> gpurun_out/r03e/bench_default.log
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03e/bench_default.json").read().strip().splitlines()[-1])
print("default", round(d["value"] / 1e6, 2), "M reads/s", round(d["ms_per_step"], 1), "ms", {k: round(v, 2) for k, v in d["kernels_ms"].items() if k != "note"}, round(d["roofline"]["frac"], 3), d["from_host"]["value"], d["from_host"]["two_contexts"]["value"])
PY
timeout -k 10 500 python3 bench.py --paired --pairs 400000 --steps 3 --warmup 1 > gpurun_out/r03e/bench_paired.json 2> gpurun_out/r03e/bench_paired.log || { tail -20 gpurun_out/r03e/bench_paired.log; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03e/bench_paired.json").read().strip().splitlines()[-1])
print("paired", round(d["value"] / 1e6, 3), "M pairs/s", round(d["ms_per_step"], 1), "ms", d["stages_s_per_step"], d["model"], d.get("cpu_baseline"))
PY
