#!/bin/bash
# Round-4 evidence for bench.py's numbers (the recipe of run_r02.sh with this round's kernels and bench flags), one gpurun call from the repo root:  bash profiles/run_r04s.sh <tag>
#   1. index built once and cached (the profiled runs only load it)
#   2. rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 5 --warmup 2 --no-cpu`  -> kernel_stats.csv
#   3. --pmc FETCH_SIZE WRITE_SIZE (own pass, counters only) over the hot-path kernels       -> pmc_hbm.csv (per kernel, per launch)
#   4. --pmc SQ issue counters (own pass)                                                   -> pmc_sq.csv
# Summaries are printed and written to gpurun_out/prof_<tag>/summary.txt; copy that directory's small files into profiles/<tag>/.
set -o pipefail
TAG=${1:-r05k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
KERNELS="ms_lf|mem_kernel|occ_kernel|pack_kernel|classify|bin_tasks|band_tasks|chain_plan|plan_kernel|dp_lane|dp_band|dp_wave|select_kernel|traceback|finish_wave|finish_kernel|finish_prep|finish_render|global_task|global_band|af_chunk|gather_lines|align_kernel"
LEAN="--no-cpu --no-from-host --no-scaling-base --no-single-context"
echo "[1/4] building + caching the index"
MONI_BENCH_SAVE_INDEX=1 python3 $ROOT/bench.py --steps 1 --warmup 0 $LEAN --inflight 1 > $OUT/bench_build.json 2> $OUT/bench_build.log || exit 1
echo "[2/4] kernel trace"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 6 --warmup 2 $LEAN > $OUT/bench_trace.json 2> $OUT/bench_trace.log || exit 1
cp "$(find $OUT/trace -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
echo "      the same with one context"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- python3 $ROOT/bench.py --steps 6 --warmup 2 $LEAN --inflight 1 > $OUT/bench_trace_inflight1.json 2> $OUT/bench_trace_inflight1.log || exit 1
cp "$(find $OUT/trace1 -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_inflight1.csv
echo "[3/4] HBM counters (FETCH_SIZE and WRITE_SIZE do not fit one pass: two passes)"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KERNELS" --output-format csv -d $OUT/pmc_hbm/fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 $LEAN --inflight 1 > $OUT/bench_pmc_fetch.json 2> $OUT/bench_pmc_fetch.log || exit 1
echo "      write pass"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KERNELS" --output-format csv -d $OUT/pmc_hbm/write -- python3 $ROOT/bench.py --steps 1 --warmup 0 $LEAN --inflight 1 > $OUT/bench_pmc_write.json 2> $OUT/bench_pmc_write.log || exit 1
echo "[4/4] SQ counters"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-include-regex "$KERNELS" --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --steps 1 --warmup 0 $LEAN --inflight 1 > $OUT/bench_pmc_sq.json 2> $OUT/bench_pmc_sq.log || exit 1
echo "[5] VALU instructions per DP cell pair: every problem through dp_lane_kernel (no bands, no dp_wave_kernel), SQ_INSTS_VALU against the cell slots the kernels count"
( export MONI_AF_DBG=196608 MONI_AF_WAVE_MAX=0; timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-include-regex "dp_lane" --output-format csv -d $OUT/pmc_dp -- python3 $ROOT/bench.py --steps 1 --warmup 0 $LEAN --inflight 1 > $OUT/bench_pmc_dp.json 2> $OUT/bench_pmc_dp.log ) || exit 1
cd $ROOT
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
def short(n):
    n = n.replace("void ", "")
    return n[:n.index("(")] if "(" in n else n
import json
for title, f, bj in (("python3 bench.py --steps 6 --warmup 2 $LEAN (two contexts in flight: 2 x 2 warm-up + 6 timed passes of 1 M reads, sub-batches of 500 k; + 3 seeding-only passes; index cached)", "$OUT/kernel_stats.csv", "$OUT/bench_trace.json"),
                     ("the same with --inflight 1 (one context: 2 + 6 passes, sub-batches of 250 k)", "$OUT/kernel_stats_inflight1.csv", "$OUT/bench_trace_inflight1.json")):
    print("== rocprofv3 --kernel-trace --stats: " + title + " ==")
    d = json.loads(open(bj).read().strip().splitlines()[-1])
    print("bench line of this run: %.2f M reads/s, %.2f ms per step; ms_lf_kernel by HIP events %.3f ms per launch, roofline.frac %.3f" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:34]:
        print("%-72s calls %6s  total %9.2f ms  avg %9.3f ms  %5.1f%%" % (short(r["Name"])[:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
    print()
def pmc(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen = set()
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])[:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (f, k, r["Dispatch_Id"]) not in seen: seen.add((f, k, r["Dispatch_Id"])); calls[k] += 1
    return agg, calls
print()
print("== --pmc FETCH_SIZE WRITE_SIZE (KB, as rocprofv3 reports them; x1024 = bytes; no x2 correction: these are 64-byte request streams), one pass of 1 M reads ==")
agg, calls = pmc("$OUT/pmc_hbm")
w = csv.writer(open("$OUT/pmc_hbm.csv", "w")); w.writerow(["kernel", "launches", "fetch_bytes_per_pass", "write_bytes_per_pass"])
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0))):
    fb, wb = v.get("FETCH_SIZE", 0) * 1024, v.get("WRITE_SIZE", 0) * 1024
    nl = calls[k] // 2 or 1          # the kernel was seen once per pass
    w.writerow([k, nl, "%.0f" % fb, "%.0f" % wb])
    print("%-60s x%-3d fetch %8.3f GB  write %8.3f GB   (per launch: %7.3f / %7.3f GB)" % (k, nl, fb / 1e9, wb / 1e9, fb / 1e9 / nl, wb / 1e9 / nl))
tf = sum(v.get("FETCH_SIZE", 0) for v in agg.values()) * 1024; tw = sum(v.get("WRITE_SIZE", 0) for v in agg.values()) * 1024
print("all hot-path kernels of the pass: fetch %.2f GB + write %.2f GB = %.2f GB" % (tf / 1e9, tw / 1e9, (tf + tw) / 1e9))
print()
print("== --pmc SQ counters per launch (profiled serially) ==")
agg, calls = pmc("$OUT/pmc_sq")
w = csv.writer(open("$OUT/pmc_sq.csv", "w")); names = sorted({c for v in agg.values() for c in v}); w.writerow(["kernel", "launches"] + names)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n = calls[k]
    w.writerow([k, n] + ["%.0f" % v.get(c, 0) for c in names])
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-60s x%-3d VALU insts %.3g  LDS insts %.3g  VALU-active/wave-cycles %.3f  WAIT_ANY/wave-cycles %.3f" % (k, n, v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_LDS", 0), v.get("SQ_ACTIVE_INST_VALU", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc))
PY
python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob, json, collections
d = json.loads(open("$OUT/bench_pmc_dp.json").read().strip().splitlines()[-1])
slots = d["align"]["roofline"]["cell_slots_run"]
valu = collections.defaultdict(float)
for f in glob.glob("$OUT/pmc_dp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "SQ_INSTS_VALU": valu[r["Kernel_Name"][:60]] += float(r["Counter_Value"])
tot = sum(valu.values())
print()
print("== VALU instructions per DP cell pair (MONI_AF_DBG=196608 MONI_AF_WAVE_MAX=0: every problem through dp_lane_kernel; one pass of 1 M reads) ==")
for k, v in sorted(valu.items(), key=lambda kv: -kv[1]): print("%-60s SQ_INSTS_VALU %.4g" % (k, v))
print("cell slots the kernels stepped through (2 problems per lane): %.4g -> lane cell pairs %.4g -> wavefront instructions per cell pair: %.4g / (%.4g / 128) = %.1f" % (slots, slots / 2, tot, slots, tot / (slots / 128)))
PY
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
