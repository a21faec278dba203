# final check of a build: the whole GPU suite, smoke, the default bench line and the paired bench line (both with their CPU baselines) -> gpurun_out/r03aq
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03aq
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03aq/pytest_gpu.log 2>&1 || { tail -60 gpurun_out/r03aq/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r03aq/pytest_gpu.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 500 python3 bench.py > gpurun_out/r03aq/bench_default.json 2> gpurun_out/r03aq/bench_default.log
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03aq/bench_default.json").read().strip().splitlines()[-1])
print("default", round(d["value"] / 1e6, 2), "M reads/s", round(d["ms_per_step"], 1), "ms; roofline", round(d["roofline"]["frac"], 3), "survey", round(d["roofline"]["survey_8d"]["frac"], 3), "cpu", round(d["cpu_baseline"]["value"]), d["cpu_baseline"]["sam_identical_on_sample"], "from_host", round(d["from_host"]["value"] / 1e6, 2), round(d["from_host"]["two_contexts"]["value"] / 1e6, 2))
PY
timeout -k 10 500 python3 bench.py --paired > gpurun_out/r03aq/bench_paired.json 2> gpurun_out/r03aq/bench_paired.log || { tail -20 gpurun_out/r03aq/bench_paired.log; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03aq/bench_paired.json").read().strip().splitlines()[-1])
print("paired", round(d["value"] / 1e6, 3), "M pairs/s", round(d["ms_per_step"], 1), "ms", d["stages_s_per_step"], "from_host", d.get("from_host"), "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("sam_identical_on_sample"), d.get("cpu_baseline", {}).get("model_identical"))
PY
