# round 3: a middle LDS instance of chain_plan_kernel as LEVEL 0 for reads of more than 200 bases (250 bp x 20 haplotypes)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03aa
timeout -k 10 900 python3 -m pytest "tests/test_gpu_fullsize.py::test_configs4_shaped_chr21_scale_20_haplotypes_250bp" tests/test_gpu_align.py -m gpu -x -q > gpurun_out/r03aa/pytest_subset.log 2>&1 || { tail -40 gpurun_out/r03aa/pytest_subset.log; exit 1; }
tail -2 gpurun_out/r03aa/pytest_subset.log
for cfg in "1 4" "1 5" "0 4"; do
  set -- $cfg
  MONI_AF_L0=$1 MONI_AF_MIDOCC=$2 timeout -k 10 400 python3 bench.py --base-len 46709983 --haps 20 --read-len 250 --steps 4 --warmup 1 --no-cpu --no-from-host > gpurun_out/r03aa/bench_250_$1_$2.json 2> gpurun_out/r03aa/bench_250_$1_$2.log
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r03aa/bench_250_$1_$2.json").read().strip().splitlines()[-1])
print("L0=$1 occ=$2:", round(d["value"] / 1e6, 2), "M reads/s", round(d["ms_per_step"], 1), "ms", d.get("handed_over_because"))
PY
done
