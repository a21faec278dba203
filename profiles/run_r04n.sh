#!/bin/bash
# round 4: dp_wave_kernel (a wavefront per pair of problems) for the queues with few problems; sub-batch size and stream count of the align stage
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04n; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== sanity: no wildcard / every read with one / one in five =="
DBG_MIX=1 timeout -k 10 300 python profiles/dbg/dbg_band.py 2>&1 | grep -v amdgpu.ids | cut -c1-260 > $OUT/sanity_0.txt || exit 1
DBG_MIX=1 DBG_N_RATE=1 timeout -k 10 300 python profiles/dbg/dbg_band.py 2>&1 | grep -v amdgpu.ids | cut -c1-260 > $OUT/sanity_1.txt || exit 1
DBG_MIX=1 DBG_N_RATE=0.2 timeout -k 10 300 python profiles/dbg/dbg_band.py 2>&1 | grep -v amdgpu.ids | cut -c1-260 > $OUT/sanity_02.txt || exit 1
cat $OUT/sanity_0.txt $OUT/sanity_1.txt $OUT/sanity_02.txt
if grep -q " [1-9][0-9]* differing" $OUT/sanity_*.txt; then echo "SANITY FAILED"; exit 1; fi
run() { # name, env...
  local name=$1; shift
  env "$@" MONI_BENCH_SAVE_INDEX=1 timeout -k 10 600 python bench.py --steps 8 --warmup 2 --no-cpu --no-from-host --no-scaling-base > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { tail -5 $OUT/bench_$name.err; return 1; }
  python - <<PY
import json; d = json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"]), round(d["ms_per_step"], 2), {k: round(v * 1e3, 2) for k, v in d["stages_s_per_step"].items()})
PY
}
echo "== bench =="
run default X=1 && run wave0 MONI_AF_WAVE_MAX=0 && run sub500k MONI_ALIGN_SUB=500000 && run sub333k MONI_ALIGN_SUB=333334 && run sub125k MONI_ALIGN_SUB=125000 && \
run nset4 MONI_HIP_LIB=$ROOT/moni_align_amd/csrc/libmoni_hip_nset4.so && run nset4_sub125k MONI_HIP_LIB=$ROOT/moni_align_amd/csrc/libmoni_hip_nset4.so MONI_ALIGN_SUB=125000 || exit 1
for nr in 0.01 0.05; do
timeout -k 10 600 python bench.py --steps 8 --warmup 2 --no-from-host --no-scaling-base --n-rate $nr --cpu-seconds 6 > $OUT/bench_nrate_$nr.json 2> $OUT/bench_nrate_$nr.err || { tail -5 $OUT/bench_nrate_$nr.err; exit 1; }
python - <<PY
import json; d = json.loads(open("$OUT/bench_nrate_$nr.json").read().strip().splitlines()[-1]); print("n-rate $nr", round(d["value"]), round(d["ms_per_step"], 2), d["align"]["reads_taken_by_general_kernel"], d["align"]["handed_over_because"], d["cpu_baseline"]["sam_identical_on_sample"])
PY
done
echo "== clean per-kernel times =="
bash profiles/clean_times.sh > $OUT/clean_times.txt 2>&1; head -26 $OUT/clean_times.txt
