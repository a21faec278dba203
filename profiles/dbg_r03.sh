cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
echo "== no huge ==" > gpurun_out/dbg/log.txt
MONI_AF_NOHUGE=1 timeout -k 10 300 python3 -m pytest tests/test_gpu_align.py -x -q -s -k "test_sam_identical_150bp" >> gpurun_out/dbg/log.txt 2>&1
echo "rc=$?" >> gpurun_out/dbg/log.txt
echo "== default ==" >> gpurun_out/dbg/log.txt
timeout -k 10 300 python3 -m pytest tests/test_gpu_align.py -x -q -s -k "test_sam_identical_150bp" >> gpurun_out/dbg/log.txt 2>&1
echo "rc=$?" >> gpurun_out/dbg/log.txt
grep -v "^  File" gpurun_out/dbg/log.txt | tail -40
