# round 3, final build: the evidence recipe of run_r02.sh (kernel trace + stats, FETCH_SIZE / WRITE_SIZE / SQ counters in their own passes) for the
# default bench, then the kernel trace + stats of the paired bench
cd $GRAFT_REPO_ROOT
bash profiles/run_r02.sh r03x > gpurun_out/run_r03x.log 2>&1 || { tail -30 gpurun_out/run_r03x.log; exit 1; }
tail -60 gpurun_out/prof_r03x/summary.txt | cut -c1-220
PAIRS=1000000 bash profiles/prof_paired.sh 2>&1 | tail -24 | cut -c1-300
