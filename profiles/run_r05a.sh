#!/bin/bash
# round 4: sharded read set (several resident chunks) with the chunks dealt to two contexts; the default line again
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05a; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
run() { local name=$1; shift
  MONI_BENCH_SAVE_INDEX=1 timeout -k 10 900 python bench.py --no-cpu --no-from-host --no-scaling-base "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { tail -8 $OUT/bench_$name.err; return 1; }
  python - <<PY
import json; d = json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"]), round(d["ms_per_step"], 2), d["scaling"], d["config"]["chunks_per_rank"], d["config"]["contexts_in_flight"], d.get("gather", {}).get("records_match_reads"), d.get("gather", {}).get("identical_to_unsharded"))
PY
}
run default --steps 6 --warmup 2 && run chunks4_inflight2 --total-reads 2000000 --reads 500000 --steps 3 --warmup 1 --gather-sam --verify-gather && run chunks4_inflight1 --total-reads 2000000 --reads 500000 --steps 3 --warmup 1 --inflight 1 && run chunks5_inflight2 --total-reads 2500000 --reads 500000 --steps 3 --warmup 1 || exit 1
echo "== scaling base leg =="
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 900 python bench.py --steps 4 --warmup 1 --no-cpu --no-from-host --no-single-context > $OUT/bench_sb.json 2> $OUT/bench_sb.err || { tail -5 $OUT/bench_sb.err; exit 1; }
python - <<PY
import json; d = json.loads(open("$OUT/bench_sb.json").read().strip().splitlines()[-1]); print("scaling_base", d["scaling_base"]["value"], d["scaling_base"]["ms_per_step"], d["scaling_base"]["contexts_in_flight"], "default", round(d["value"]))
PY
