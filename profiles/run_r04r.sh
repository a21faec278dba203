#!/bin/bash
# round 4: -c for pairs (library and binary) against the oracle; address translation and L1->L2 latency counters of ms_lf_kernel / mem_kernel / occ_kernel
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04r; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
echo "== -c for pairs =="
timeout -k 10 900 python -m pytest tests/test_gpu_pe.py tests/test_cli.py -m gpu -x -q -k "csv or paired_end or unsupported" > $OUT/pytest_csv.log 2>&1; rc=$?; tail -5 $OUT/pytest_csv.log
[ $rc -ne 0 ] && { grep -a -B5 -A25 "Error" $OUT/pytest_csv.log | head -80 | cut -c1-600; exit $rc; }
echo "== translation counters =="
MONI_PMC_CTRS="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" bash profiles/pmc_kernel.sh r04r_utcl "ms_lf_kernel|mem_kernel|occ_kernel" "X=1" 2>&1 | tail -8
echo "== L1 -> L2 read requests and their latency =="
MONI_PMC_CTRS="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" bash profiles/pmc_kernel.sh r04r_lat "ms_lf_kernel|mem_kernel|occ_kernel" "X=1" 2>&1 | tail -8
