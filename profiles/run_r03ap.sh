# round 3: the 192-anchor middle instance as the default LEVEL 0 for reads of more than 200 bases: parity at 250 bp x 20 haplotypes, the bench line
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03ap
timeout -k 10 900 python3 -m pytest "tests/test_gpu_fullsize.py::test_configs4_shaped_chr21_scale_20_haplotypes_250bp" tests/test_gpu_align.py tests/test_gpu_pe.py -m gpu -x -q > gpurun_out/r03ap/pytest_subset.log 2>&1 || { tail -40 gpurun_out/r03ap/pytest_subset.log; exit 1; }
tail -2 gpurun_out/r03ap/pytest_subset.log
timeout -k 10 500 python3 bench.py --base-len 46709983 --haps 20 --read-len 250 --steps 5 --warmup 1 --no-from-host > gpurun_out/r03ap/bench_config5_250bp_20hap.json 2> gpurun_out/r03ap/bench_250.log
python3 - <<PY
import json
d = json.loads(open("gpurun_out/r03ap/bench_config5_250bp_20hap.json").read().strip().splitlines()[-1])
print("250 bp x 20 haplotypes:", round(d["value"] / 1e6, 3), "M reads/s", round(d["ms_per_step"], 1), "ms", d.get("handed_over_because"), d["cpu_baseline"]["sam_identical_on_sample"], round(d["cpu_baseline"]["value"]))
PY
