set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03w
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
MONI_AK_PROFILE=1 timeout -k 10 500 python3 bench.py --paired --steps 3 --warmup 1 --no-cpu --no-from-host > gpurun_out/r03w/bench_paired.json 2> gpurun_out/r03w/bench_paired.log
grep "paired path\|paired batch" gpurun_out/r03w/bench_paired.log | tail -4
