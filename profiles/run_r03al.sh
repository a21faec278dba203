# round 3: the local alignment with two target rows per lane
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03al
timeout -k 10 900 python3 -m pytest tests/test_gpu_pe.py tests/test_golden.py "tests/test_gpu_fullsize.py::test_configs2_paired_end_with_orphan_recovery" -m gpu -x -q > gpurun_out/r03al/pytest_subset.log 2>&1 || { tail -60 gpurun_out/r03al/pytest_subset.log; exit 1; }
tail -2 gpurun_out/r03al/pytest_subset.log
timeout -k 10 500 python3 bench.py --paired --steps 5 --warmup 1 --no-from-host > gpurun_out/r03al/bench_paired.json 2> gpurun_out/r03al/bench_paired.log
python3 - <<PY
import json
d = json.loads(open("gpurun_out/r03al/bench_paired.json").read().strip().splitlines()[-1])
print(round(d["value"] / 1e6, 3), "M pairs/s", round(d["ms_per_step"], 1), "ms", d.get("cpu_baseline", {}).get("sam_identical_on_sample"))
PY
bash profiles/pe_timeline.sh > gpurun_out/r03al/timeline.txt 2>&1; grep -E "pe_orphan|step span" gpurun_out/r03al/timeline.txt | tail -14
