# round 3: a 192-anchor middle instance against the 160-anchor one at 250 bp x 20 haplotypes
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03ao
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --base-len 46709983 --haps 20 --read-len 250 --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
for mid in 1 2; do
  MONI_AF_MID=$mid timeout -k 10 400 python3 bench.py --base-len 46709983 --haps 20 --read-len 250 --steps 5 --warmup 1 --no-cpu --no-from-host > gpurun_out/r03ao/b_$mid.json 2> gpurun_out/r03ao/b_$mid.log
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r03ao/b_$mid.json").read().strip().splitlines()[-1])
print("middle instance $mid:", round(d["value"] / 1e6, 3), "M reads/s", round(d["ms_per_step"], 1), "ms", d.get("handed_over_because"))
PY
done | tee gpurun_out/r03ao/sweep.txt
