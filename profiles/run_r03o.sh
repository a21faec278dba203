# round 3: hardware queues for the paired path's streams
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03o
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
for q in 4 8 16; do
  export GPU_MAX_HW_QUEUES=$q
  MONI_AK_PROFILE=1 timeout -k 10 500 python3 bench.py --paired --pairs 1000000 --steps 4 --warmup 1 --no-cpu > gpurun_out/r03o/bench_paired_q$q.json 2> gpurun_out/r03o/bench_paired_q$q.log || { tail -20 gpurun_out/r03o/bench_paired_q$q.log; exit 1; }
  echo "queues $q"; grep "paired batch" gpurun_out/r03o/bench_paired_q$q.log | tail -2
done
