# round 3: paired bench with the mates resident (moni_pe_align_run), a small last chunk; from_host beside it
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03r
timeout -k 10 600 python3 -m pytest tests/test_gpu_pe.py tests/test_golden.py -m gpu -x -q > gpurun_out/r03r/pytest_subset.log 2>&1 || { tail -60 gpurun_out/r03r/pytest_subset.log; exit 1; }
tail -2 gpurun_out/r03r/pytest_subset.log
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
for even in 0 1; do
  if [ $even = 1 ]; then export MONI_PE_EVEN_CHUNKS=1; fi
  MONI_AK_PROFILE=1 timeout -k 10 500 python3 bench.py --paired --steps 4 --warmup 1 --no-cpu > gpurun_out/r03r/bench_paired_even$even.json 2> gpurun_out/r03r/bench_paired_even$even.log || { tail -20 gpurun_out/r03r/bench_paired_even$even.log; exit 1; }
  grep "paired batch" gpurun_out/r03r/bench_paired_even$even.log | tail -2
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r03r/bench_paired_even$even.json").read().strip().splitlines()[-1])
print("paired even=$even", round(d["value"] / 1e6, 3), "M pairs/s", round(d["ms_per_step"], 1), "ms", d["stages_s_per_step"], d.get("from_host"), d.get("pairs_handed_over"), d.get("handed_over_because"))
PY
done
