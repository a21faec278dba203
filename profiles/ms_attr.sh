# attribution of ms_lf_kernel's requests beyond one per step.  Build the two variants first (in moni_align_amd/csrc):
#   for v in 1 2; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMONI_MS_ATTR=$v -o libmoni_hip_attr$v.so moni_hip.hip -Wl,pe_big.o; done
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ms_attr
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host > /dev/null 2>&1
for v in 1 2; do
  MONI_HIP_LIB=$GRAFT_REPO_ROOT/moni_align_amd/csrc/libmoni_hip_attr$v.so timeout -k 10 300 python3 profiles/ms_attr.py 2>&1 | tail -1
done | tee gpurun_out/ms_attr/attr.txt
