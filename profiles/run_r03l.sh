# round 3: what pe_plan_kernel's LDS instance has to hold (seed / anchor counts per pair)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03l
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
timeout -k 10 300 python3 profiles/pe_seed_hist.py > gpurun_out/r03l/pe_seed_hist.txt 2>&1 || { tail -20 gpurun_out/r03l/pe_seed_hist.txt; exit 1; }
cat gpurun_out/r03l/pe_seed_hist.txt
