#!/bin/bash
# round 4, final build (bash profiles/run_r05c.sh [tag]): the whole GPU suite, smoke, the driver's bench command, the robustness leg, the other configurations, the paired lines
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r05c}; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== GPU tests =="
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -3 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && { grep -a -B5 -A25 "Error" $OUT/pytest_gpu.log | head -80 | cut -c1-500; exit $rc; }
echo "== smoke =="
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
x = {"value": round(d["value"]), "ms": round(d["ms_per_step"], 2)}
if "roofline" in d: x["frac"] = round(d["roofline"]["frac"], 3); x["alone"] = round(d["roofline"].get("kernel_alone", {}).get("frac", 0), 3)
if "single_context" in d: x["single_ms"] = round(d["single_context"]["ms_per_step"], 2)
if "cpu_baseline" in d: x["cpu"] = round(d["cpu_baseline"]["value"]); x["sam_identical"] = d["cpu_baseline"].get("sam_identical_on_sample")
if "scaling_base" in d: x["scaling_base"] = round(d["scaling_base"]["value"])
if "from_host" in d and d["from_host"]: x["from_host"] = round(d["from_host"]["value"]); x["from_host_2ctx"] = round(d["from_host"].get("two_contexts", {}).get("value", 0))
if "device_memory_GB" in d: x["mem_GB"] = round(d["device_memory_GB"]["used_after_timed_region"], 1)
if "align" in d: x["to_align_kernel"] = d["align"]["reads_taken_by_general_kernel"]; x["why"] = d["align"]["handed_over_because"]; x["align_frac"] = round(d["align"]["roofline"]["frac"], 3)
if "pairs_taken_by_pe_align_kernel" in d: x["to_pe_align_kernel"] = d["pairs_taken_by_pe_align_kernel"]
print(sys.argv[1], x)
PY
}
echo "== the driver's command =="
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err || { tail -5 $OUT/bench_driver.err; exit 1; }
line driver $OUT/bench_driver.json
echo "== 1 % of the reads with an N =="
timeout -k 10 600 python bench.py --steps 10 --warmup 2 --no-from-host --no-scaling-base --n-rate 0.01 --cpu-seconds 6 > $OUT/bench_nrate_0.01.json 2> $OUT/bench_nrate_0.01.err || { tail -5 $OUT/bench_nrate_0.01.err; exit 1; }
line nrate_0.01 $OUT/bench_nrate_0.01.json
echo "== paired =="
timeout -k 10 900 python bench.py --paired --steps 6 --warmup 1 --cpu-seconds 6 > $OUT/bench_paired.json 2> $OUT/bench_paired.err || { tail -5 $OUT/bench_paired.err; exit 1; }
line paired $OUT/bench_paired.json
timeout -k 10 900 python bench.py --paired -Z --steps 4 --warmup 1 --no-cpu --no-from-host > $OUT/bench_paired_Z.json 2> $OUT/bench_paired_Z.err || { tail -5 $OUT/bench_paired_Z.err; exit 1; }
line paired_Z $OUT/bench_paired_Z.json
echo "== repeat-rich index; 250 bp x 20 haplotypes =="
timeout -k 10 900 python bench.py --steps 6 --warmup 2 --no-from-host --no-scaling-base --no-cpu --repeats 0.05 > $OUT/bench_repeats_0.05.json 2> $OUT/bench_repeats_0.05.err || { tail -5 $OUT/bench_repeats_0.05.err; exit 1; }
line repeats $OUT/bench_repeats_0.05.json
timeout -k 10 900 python bench.py --steps 6 --warmup 2 --no-from-host --no-scaling-base --cpu-seconds 6 --base-len 46709983 --haps 20 --read-len 250 > $OUT/bench_config5_250bp_20hap.json 2> $OUT/bench_config5_250bp_20hap.err || { tail -5 $OUT/bench_config5_250bp_20hap.err; exit 1; }
line 250bp_20hap $OUT/bench_config5_250bp_20hap.json
