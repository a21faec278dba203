# round 3: the scoring pass of orphan recovery on up to 4096 waves
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03v
timeout -k 10 900 python3 -m pytest tests/test_gpu_pe.py tests/test_golden.py -m gpu -x -q > gpurun_out/r03v/pytest_subset.log 2>&1 || { tail -60 gpurun_out/r03v/pytest_subset.log; exit 1; }
tail -2 gpurun_out/r03v/pytest_subset.log
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
for cfg in "8 4096" "8 2048" "4 4096" "16 4096"; do
  set -- $cfg
  MONI_PE_NSPLIT=$1 MONI_PE_K1_WAVES=$2 MONI_AK_PROFILE=1 timeout -k 10 500 python3 bench.py --paired --steps 4 --warmup 1 --no-cpu --no-from-host > gpurun_out/r03v/bench_paired_$1_$2.json 2> gpurun_out/r03v/bench_paired_$1_$2.log || { tail -20 gpurun_out/r03v/bench_paired_$1_$2.log; exit 1; }
  echo "nsplit $1 waves $2"; grep "paired batch" gpurun_out/r03v/bench_paired_$1_$2.log | tail -1
done
bash profiles/pe_timeline.sh > gpurun_out/r03v/timeline.txt 2>&1; grep -E "pe_orphan|pe_lines|step span" gpurun_out/r03v/timeline.txt | tail -20
