#!/bin/bash
# SQ counters of finish_prep_kernel cut short (-DAF_CUTS build)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash profiles/pmc_kernel.sh r04y "finish_prep" "MONI_HIP_LIB=$ROOT/moni_align_amd/csrc/libmoni_hip_cuts.so MONI_AF_DBG=1048576" "MONI_HIP_LIB=$ROOT/moni_align_amd/csrc/libmoni_hip_cuts.so MONI_AF_DBG=4194304" "MONI_HIP_LIB=$ROOT/moni_align_amd/csrc/libmoni_hip_cuts.so MONI_AF_DBG=0" 2>&1 | grep -E "^==|finish_prep" | cut -c1-330
