# round 3: memo of solved DP requests in the orphan scoring pass (chains in blocks, not strided)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03ad
timeout -k 10 900 python3 -m pytest tests/test_gpu_pe.py tests/test_golden.py "tests/test_gpu_fullsize.py::test_configs2_paired_end_with_orphan_recovery" -m gpu -x -q > gpurun_out/r03ad/pytest_subset.log 2>&1 || { tail -60 gpurun_out/r03ad/pytest_subset.log; exit 1; }
tail -2 gpurun_out/r03ad/pytest_subset.log
for ns in 8 4 16; do
  MONI_PE_NSPLIT=$ns MONI_AK_PROFILE=1 timeout -k 10 500 python3 bench.py --paired --steps 4 --warmup 1 --no-from-host > gpurun_out/r03ad/bench_paired_ns$ns.json 2> gpurun_out/r03ad/bench_paired_ns$ns.log || { tail -20 gpurun_out/r03ad/bench_paired_ns$ns.log; exit 1; }
  echo "nsplit $ns"; grep "paired batch" gpurun_out/r03ad/bench_paired_ns$ns.log | tail -1
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r03ad/bench_paired_ns$ns.json").read().strip().splitlines()[-1])
print("  ", round(d["value"] / 1e6, 3), "M pairs/s", d.get("cpu_baseline", {}).get("sam_identical_on_sample"), d["dp_problems"], d["dp_cells"])
PY
done
bash profiles/pe_timeline.sh > gpurun_out/r03ad/timeline.txt 2>&1; grep -E "pe_orphan|step span" gpurun_out/r03ad/timeline.txt | tail -14
