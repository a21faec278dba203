# round 3: orphan recovery's chain loop over several waves (pe_orphan_kernel: park / score in parts / replay)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03u
timeout -k 10 900 python3 -m pytest tests/test_gpu_pe.py tests/test_golden.py "tests/test_gpu_fullsize.py::test_configs2_paired_end_with_orphan_recovery" -m gpu -x -q > gpurun_out/r03u/pytest_subset.log 2>&1 || { tail -60 gpurun_out/r03u/pytest_subset.log; exit 1; }
tail -2 gpurun_out/r03u/pytest_subset.log
for ns in 8 0 16; do
  MONI_PE_NSPLIT=$ns MONI_AK_PROFILE=1 timeout -k 10 500 python3 bench.py --paired --steps 4 --warmup 1 --no-cpu --no-from-host > gpurun_out/r03u/bench_paired_ns$ns.json 2> gpurun_out/r03u/bench_paired_ns$ns.log || { tail -20 gpurun_out/r03u/bench_paired_ns$ns.log; exit 1; }
  echo "nsplit $ns"; grep "paired batch" gpurun_out/r03u/bench_paired_ns$ns.log | tail -2
done
bash profiles/pe_timeline.sh > gpurun_out/r03u/timeline.txt 2>&1; grep -E "pe_orphan|pe_align|pe_lines|step span|copyBuffer" gpurun_out/r03u/timeline.txt | tail -30
