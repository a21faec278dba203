# round 3: pe_plan_kernel in two levels (small LDS instance at 4 waves/SIMD, the large one for pairs that overflow it)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03m
timeout -k 10 900 python3 -m pytest tests/test_gpu_pe.py tests/test_golden.py tests/test_cli.py -m gpu -x -q > gpurun_out/r03m/pytest_subset.log 2>&1 || { tail -60 gpurun_out/r03m/pytest_subset.log; exit 1; }
tail -2 gpurun_out/r03m/pytest_subset.log
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
for mode in two one; do
  if [ $mode = one ]; then export MONI_PE_ONE_LEVEL=1; fi
  MONI_AK_PROFILE=1 timeout -k 10 500 python3 bench.py --paired --pairs 1000000 --steps 3 --warmup 1 --no-cpu > gpurun_out/r03m/bench_paired_$mode.json 2> gpurun_out/r03m/bench_paired_$mode.log || { tail -20 gpurun_out/r03m/bench_paired_$mode.log; exit 1; }
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r03m/bench_paired_$mode.json").read().strip().splitlines()[-1])
print("paired $mode", round(d["value"] / 1e6, 3), "M pairs/s", round(d["ms_per_step"], 1), "ms", d["stages_s_per_step"], d["pairs_through_host_pipeline"])
PY
done
unset MONI_PE_ONE_LEVEL
PAIRS=1000000 bash profiles/prof_paired.sh 2>&1 | tail -18
