# round 3: 250 bp x 20 haplotypes against the finish grid and the hardware queues
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03ar
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --base-len 46709983 --haps 20 --read-len 250 --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
for cfg in "96 16" "48 16" "192 16" "96 32"; do
  set -- $cfg
  MONI_AF_FINGRID=$1 GPU_MAX_HW_QUEUES=$2 timeout -k 10 400 python3 bench.py --base-len 46709983 --haps 20 --read-len 250 --steps 5 --warmup 1 --no-cpu --no-from-host > gpurun_out/r03ar/b_$1_$2.json 2> gpurun_out/r03ar/b_$1_$2.log
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r03ar/b_$1_$2.json").read().strip().splitlines()[-1])
print("fingrid $1 queues $2:", round(d["value"] / 1e6, 3), "M reads/s", round(d["ms_per_step"], 1), "ms")
PY
done | tee gpurun_out/r03ar/sweep.txt
