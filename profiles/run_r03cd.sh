# round 3 (r03c, r03d): parity subset, then the three bench configurations (default, 5 % repeats, 250 bp x 20 haplotypes)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
timeout -k 10 1000 python3 -m pytest tests/test_ms_index_io.py tests/test_gpu_seed.py tests/test_gpu_align.py tests/test_golden.py tests/test_gpu_fullsize.py tests/test_cli.py -m gpu -x -q > gpurun_out/r03d/pytest_subset.log 2>&1 || { tail -40 gpurun_out/r03d/pytest_subset.log; exit 1; }
tail -2 gpurun_out/r03d/pytest_subset.log
timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/r03d/bench_default.json 2> gpurun_out/r03d/bench_default.log
timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-cpu --no-from-host --repeats 0.05 > gpurun_out/r03d/bench_repeats_0.05.json 2> gpurun_out/r03d/bench_repeats_0.05.log
timeout -k 10 500 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-from-host --base-len 46709983 --haps 20 --read-len 250 > gpurun_out/r03d/bench_config5_250bp_20hap.json 2> gpurun_out/r03d/bench_config5_250bp_20hap.log
python3 - <<'PY'
import json
for f in ("bench_default", "bench_repeats_0.05", "bench_config5_250bp_20hap"):
    d = json.loads(open("gpurun_out/r03d/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"] / 1e6, 2), "M reads/s", round(d["ms_per_step"], 1), "ms", d["align"]["reads_taken_by_general_kernel"], d["align"]["reads_handed_to_host_pipeline"], d["align"]["handed_over_because"], d["align"]["kernels_ms_per_step_summed"], d["stages_s_per_step"])
PY
