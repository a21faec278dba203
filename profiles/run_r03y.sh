# round 3: smoke(), the 2-rank rehearsals (single-end: bench.py launches its own ranks, 10 M-read style sharding scaled down; paired: model learnt on rank 0
# and broadcast), and the timeline of a 250 bp x 20 haplotypes step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03y
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
MONI_BENCH_BACKEND=gloo MONI_BENCH_DEVICE=0 timeout -k 10 400 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu --total-reads 2000000 --verify-gather > gpurun_out/r03y/rehearse_2ranks.json 2> gpurun_out/r03y/rehearse_2ranks.log || { tail -20 gpurun_out/r03y/rehearse_2ranks.log; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03y/rehearse_2ranks.json").read().strip().splitlines()[-1])
print("2 ranks SE:", {k: d.get(k) for k in ("value", "n_gpus", "scaling", "ms_per_step")}, d.get("gather"), d["config"].get("launched_by"))
PY
MONI_BENCH_BACKEND=gloo MONI_BENCH_DEVICE=0 timeout -k 10 400 python3 bench.py --paired --pairs 400000 --gpus 2 --steps 2 --warmup 1 --no-cpu --no-from-host > gpurun_out/r03y/rehearse_2ranks_paired.json 2> gpurun_out/r03y/rehearse_2ranks_paired.log || { tail -20 gpurun_out/r03y/rehearse_2ranks_paired.log; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03y/rehearse_2ranks_paired.json").read().strip().splitlines()[-1])
print("2 ranks PE:", {k: d.get(k) for k in ("value", "n_gpus", "scaling", "ms_per_step", "aligned_pairs_all_ranks", "model")})
PY
