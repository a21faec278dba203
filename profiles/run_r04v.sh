#!/bin/bash
# counters of finish_prep_kernel / finish_render_kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash profiles/pmc_kernel.sh r04v "finish_prep|finish_render" "X=1" 2>&1 | tail -4
MONI_PMC_CTRS="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" bash profiles/pmc_kernel.sh r04v_lat "finish_prep|finish_render" "X=1" 2>&1 | tail -3
MONI_PMC_CTRS="TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum" bash profiles/pmc_kernel.sh r04v_wr "finish_prep|finish_render" "X=1" 2>&1 | tail -3
MONI_PMC_CTRS="FETCH_SIZE" bash profiles/pmc_kernel.sh r04v_f "finish_prep|finish_render" "X=1" 2>&1 | tail -3
MONI_PMC_CTRS="WRITE_SIZE" bash profiles/pmc_kernel.sh r04v_w "finish_prep|finish_render" "X=1" 2>&1 | tail -3
