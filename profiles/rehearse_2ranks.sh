set -e
cd $GRAFT_REPO_ROOT
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu > gpurun_out/reh_build.json 2> gpurun_out/reh_build.log
MONI_BENCH_BACKEND=gloo MONI_BENCH_DEVICE=0 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu > gpurun_out/reh2.json 2> gpurun_out/reh2.log
tail -3 gpurun_out/reh2.log
grep '"metric"' gpurun_out/reh2.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','n_gpus','ms_per_step','aligned_all_ranks','host')}); print(d['config']['parallelism'], d['stages_s_per_step'])"
