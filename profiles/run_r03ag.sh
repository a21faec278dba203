set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03ag
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
for last in 3 8 12 24; do
  MONI_PE_NSPLIT_LAST=$last timeout -k 10 500 python3 bench.py --paired --steps 5 --warmup 1 --no-cpu --no-from-host > gpurun_out/r03ag/b_$last.json 2> gpurun_out/r03ag/b_$last.log
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r03ag/b_$last.json").read().strip().splitlines()[-1])
print("parts in the last chunk $last:", round(d["value"] / 1e6, 3), "M pairs/s", round(d["ms_per_step"], 1), "ms")
PY
done | tee gpurun_out/r03ag/sweep.txt
timeout -k 10 300 python3 -m pytest tests/test_gpu_pe.py -m gpu -x -q 2>&1 | tail -2
