# round 3: the two lines of every pair written and ordered on the GPU (pe_lines_kernel)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03k
timeout -k 10 900 python3 -m pytest tests/test_gpu_pe.py tests/test_golden.py tests/test_cli.py -m gpu -x -q > gpurun_out/r03k/pytest_subset.log 2>&1 || { tail -60 gpurun_out/r03k/pytest_subset.log; exit 1; }
tail -2 gpurun_out/r03k/pytest_subset.log
MONI_AK_PROFILE=1 timeout -k 10 500 python3 bench.py --paired --pairs 1000000 --steps 3 --warmup 1 --no-cpu > gpurun_out/r03k/bench_paired.json 2> gpurun_out/r03k/bench_paired.log || { tail -20 gpurun_out/r03k/bench_paired.log; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03k/bench_paired.json").read().strip().splitlines()[-1])
print("paired", round(d["value"] / 1e6, 3), "M pairs/s", round(d["ms_per_step"], 1), "ms", d["stages_s_per_step"], d["pairs_through_host_pipeline"])
PY
PAIRS=1000000 bash profiles/prof_paired.sh 2>&1 | tail -16
