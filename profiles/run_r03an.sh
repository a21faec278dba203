# round 3: pairs per chunk and parts per parked pair once more on the final kernels
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03an
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
for cfg in "262144 3" "349526 3" "524288 3" "262144 2" "262144 4" "200000 3"; do
  set -- $cfg
  MONI_PE_CHUNK=$1 MONI_PE_NSPLIT=$2 timeout -k 10 500 python3 bench.py --paired --steps 5 --warmup 1 --no-cpu --no-from-host > gpurun_out/r03an/b_$1_$2.json 2> gpurun_out/r03an/b_$1_$2.log
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r03an/b_$1_$2.json").read().strip().splitlines()[-1])
print("chunk $1 parts $2:", round(d["value"] / 1e6, 3), "M pairs/s", round(d["ms_per_step"], 1), "ms")
PY
done | tee gpurun_out/r03an/sweep.txt
