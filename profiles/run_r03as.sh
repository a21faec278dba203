# round 3: last look at bench.py's launcher paths after the cache change (2 ranks on the one GPU over gloo, no cache file beforehand)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03as
rm -rf /tmp/moni_bench_cache
MONI_BENCH_BACKEND=gloo MONI_BENCH_DEVICE=0 timeout -k 10 500 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu --total-reads 2000000 > gpurun_out/r03as/rehearse_2ranks.json 2> gpurun_out/r03as/rehearse_2ranks.log || { tail -20 gpurun_out/r03as/rehearse_2ranks.log; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03as/rehearse_2ranks.json").read().strip().splitlines()[-1])
print("2 ranks:", {k: d.get(k) for k in ("value", "n_gpus", "scaling", "ms_per_step")}, d.get("gather", {}).get("records_match_reads"), d["roofline"]["frac"])
PY
