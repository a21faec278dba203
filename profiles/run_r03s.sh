# round 3: paired-end at full size against the oracle (table of the pairing term from the host's libm), the other paired tests
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03s
timeout -k 10 1100 python3 -m pytest tests/test_gpu_pe.py tests/test_golden.py "tests/test_gpu_fullsize.py::test_configs2_paired_end_with_orphan_recovery" -m gpu -x -q > gpurun_out/r03s/pytest_subset.log 2>&1 || { tail -60 gpurun_out/r03s/pytest_subset.log; exit 1; }
tail -3 gpurun_out/r03s/pytest_subset.log
