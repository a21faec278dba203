# round 3, first GPU call: parity suite, the default bench line, the 2-rank rehearsal of the self-launching bench (two ranks share the
# box's one GPU: gloo carries the collectives, RCCL refuses two ranks on one device)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/r03a/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/r03a/pytest_gpu.log
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 500 python3 bench.py --steps 10 --warmup 3 > gpurun_out/r03a/bench_default.json 2> gpurun_out/r03a/bench_default.log
python3 -c "
import json; d=json.loads(open('gpurun_out/r03a/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling')}); print(d['from_host']); print(d['cpu_baseline'])"
MONI_BENCH_BACKEND=gloo MONI_BENCH_DEVICE=0 timeout -k 10 400 python3 bench.py --gpus 2 --total-reads 2500000 --verify-gather --no-cpu --steps 3 --warmup 1 > gpurun_out/r03a/rehearse_2ranks.json 2> gpurun_out/r03a/rehearse_2ranks.log
python3 -c "
import json; d=json.loads(open('gpurun_out/r03a/rehearse_2ranks.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling','value_with_gather')}); print(d['gather']); print(d['config']['launched_by'], d['config']['chunks_per_rank'])"
