# round 3: the paired front end (reader thread, two workers per GPU, blocks written in place): CLI tests, then 2 M pairs from FASTQ files to a SAM file
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03z
true
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
timeout -k 10 700 python3 profiles/pe_frontend.py > gpurun_out/r03z/pe_frontend.log 2>&1 || { tail -30 gpurun_out/r03z/pe_frontend.log; exit 1; }
cat gpurun_out/r03z/pe_frontend.log
