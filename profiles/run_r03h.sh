# round 3: staged paired-end kernels with the hand-over pairs one per wave on their own stream: parity, the paired bench line, its kernel trace
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03h
timeout -k 10 900 python3 -m pytest tests/test_gpu_pe.py tests/test_golden.py tests/test_gpu_seed.py tests/test_cli.py -m gpu -x -q > gpurun_out/r03h/pytest_subset.log 2>&1 || { tail -60 gpurun_out/r03h/pytest_subset.log; exit 1; }
tail -2 gpurun_out/r03h/pytest_subset.log
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-cpu --no-from-host > gpurun_out/r03h/bench_default.json 2> gpurun_out/r03h/bench_default.log
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03h/bench_default.json").read().strip().splitlines()[-1])
print("default", round(d["value"] / 1e6, 2), "M reads/s", round(d["ms_per_step"], 1), "ms", {k: round(v, 2) for k, v in d["kernels_ms"].items() if k != "note"}, round(d["roofline"]["frac"], 3))
PY
timeout -k 10 500 python3 bench.py --paired --pairs 1000000 --steps 3 --warmup 1 > gpurun_out/r03h/bench_paired.json 2> gpurun_out/r03h/bench_paired.log || { tail -20 gpurun_out/r03h/bench_paired.log; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03h/bench_paired.json").read().strip().splitlines()[-1])
print("paired", round(d["value"] / 1e6, 3), "M pairs/s", round(d["ms_per_step"], 1), "ms", d["stages_s_per_step"], d["pairs_through_host_pipeline"], d.get("cpu_baseline"))
PY
PAIRS=1000000 bash profiles/prof_paired.sh 2>&1 | tail -20
