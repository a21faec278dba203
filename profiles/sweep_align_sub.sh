set -e
python -m pytest tests/test_gpu_align.py -x -q > gpurun_out/t.log 2>&1 || { tail -20 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
for s in 32768 50000 65536 100000 200000; do
  MONI_ALIGN_SUB=$s MONI_AK_PROFILE=1 python bench.py --no-cpu --steps 2 --warmup 1 > gpurun_out/b_$s.log 2>gpurun_out/b_$s.err
  python -c "import json,sys;d=json.loads(open('gpurun_out/b_$s.log').read().strip().splitlines()[-1]);f=d['full_path'];print($s, round(f['value']), f['seconds'])"
done
