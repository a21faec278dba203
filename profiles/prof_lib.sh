#!/bin/bash
# one bench step with the AF_PROFILE build of the library (in-kernel phase stamps), from the repo root through gpurun
MONI_HIP_LIB=$PWD/moni_align_amd/csrc/libmoni_hip_prof.so MONI_AK_PROFILE=1 python3 bench.py --steps 1 --warmup 1 --no-cpu "$@" 2>&1 >/dev/null | grep "wave cycles (set\|staged kernels" | tail -4
