# round 3: fast rows indexed by symbol (a step reads one 32-byte entry, issued at once): parity, default bench, ms_lf request counters
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
timeout -k 10 600 python3 -m pytest tests/test_gpu_seed.py tests/test_golden.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r03f/pytest_subset.log 2>&1 || { tail -40 gpurun_out/r03f/pytest_subset.log; exit 1; }
tail -2 gpurun_out/r03f/pytest_subset.log
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/r03f/bench_default.json 2> gpurun_out/r03f/bench_default.log
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03f/bench_default.json").read().strip().splitlines()[-1])
print("default", round(d["value"] / 1e6, 2), "M reads/s", round(d["ms_per_step"], 1), "ms", {k: round(v, 2) for k, v in d["kernels_ms"].items() if k != "note"}, round(d["roofline"]["frac"], 3), d["from_host"]["value"], d["from_host"]["two_contexts"]["value"])
PY
bash profiles/pmc_mslf.sh 2>&1 | tail -9
