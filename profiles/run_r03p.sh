# round 3: pairs per chunk (the DP kernels of a 65 k-pair chunk leave most wave slots empty: ~700 chunks of 128 problems for 3072 slots)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03p
export GPU_MAX_HW_QUEUES=16
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
for ch in 131072 262144 524288; do
  export MONI_PE_CHUNK=$ch
  MONI_AK_PROFILE=1 timeout -k 10 500 python3 bench.py --paired --pairs 1000000 --steps 4 --warmup 1 --no-cpu > gpurun_out/r03p/bench_paired_c$ch.json 2> gpurun_out/r03p/bench_paired_c$ch.log || { tail -20 gpurun_out/r03p/bench_paired_c$ch.log; exit 1; }
  echo "chunk $ch"; grep "paired batch" gpurun_out/r03p/bench_paired_c$ch.log | tail -2
done
