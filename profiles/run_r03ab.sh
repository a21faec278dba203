# round 3: 250 bp x 20 haplotypes against the sub-batch size
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03ab
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --base-len 46709983 --haps 20 --read-len 250 --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
for sub in 250000 333334 500000 200000; do
  MONI_ALIGN_SUB=$sub timeout -k 10 400 python3 bench.py --base-len 46709983 --haps 20 --read-len 250 --steps 4 --warmup 1 --no-cpu --no-from-host > gpurun_out/r03ab/bench_250_s$sub.json 2> gpurun_out/r03ab/bench_250_s$sub.log
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r03ab/bench_250_s$sub.json").read().strip().splitlines()[-1])
print("sub $sub:", round(d["value"] / 1e6, 2), "M reads/s", round(d["ms_per_step"], 1), "ms", {k: round(v, 1) for k, v in d["kernels_ms"].items() if k != "note"})
PY
done
