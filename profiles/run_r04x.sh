#!/bin/bash
# finish_prep_kernel cut short phase by phase (-DAF_CUTS build): clean per-kernel times
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for v in 1048576 2097152 4194304 0; do
  echo "== MONI_AF_DBG=$v =="
  MONI_AF_DBG=$v MONI_HIP_LIB=$ROOT/moni_align_amd/csrc/libmoni_hip_cuts.so bash profiles/clean_times.sh 2>&1 | grep -E "finish_prep|finish_render"
done
