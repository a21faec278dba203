# round 3: single-end step against sub-batch size and hardware queues (the lines are ordered on the GPU now; round 2's sweep had a host stage per sub-batch)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03t
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
for q in 8 16; do for sub in 200000 250000 333334 500000; do
  GPU_MAX_HW_QUEUES=$q MONI_ALIGN_SUB=$sub timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu --no-from-host > gpurun_out/r03t/b_q${q}_s$sub.json 2> gpurun_out/r03t/b_q${q}_s$sub.log
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r03t/b_q${q}_s$sub.json").read().strip().splitlines()[-1])
print("queues $q sub $sub:", round(d["value"] / 1e6, 2), "M reads/s", round(d["ms_per_step"], 2), "ms")
PY
done; done | tee gpurun_out/r03t/sweep.txt
