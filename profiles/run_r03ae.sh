# round 3: parts per parked pair x waves of the scoring pass, with the memo
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03ae
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
for cfg in "2 4096" "3 4096" "4 4096" "4 1024" "6 4096" "4 2048"; do
  set -- $cfg
  MONI_PE_NSPLIT=$1 MONI_PE_K1_WAVES=$2 timeout -k 10 500 python3 bench.py --paired --steps 5 --warmup 1 --no-cpu --no-from-host > gpurun_out/r03ae/b_$1_$2.json 2> gpurun_out/r03ae/b_$1_$2.log
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r03ae/b_$1_$2.json").read().strip().splitlines()[-1])
print("nsplit $1 waves $2:", round(d["value"] / 1e6, 3), "M pairs/s", round(d["ms_per_step"], 1), "ms")
PY
done | tee gpurun_out/r03ae/sweep.txt
