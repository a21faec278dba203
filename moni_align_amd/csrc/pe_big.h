// Host pipeline for the pairs that exceed pe_align_kernel's capacities (the paired counterpart of the single-end host pipeline): the
// same per-pair state machine (pe_core.h, compiled a second time with large capacities in pe_big.cpp) driven on the host, its DP
// problems solved on the GPU in batches (dp_run).  Interface between the two translation units: plain data only.
#pragma once
#include <functional>
#include <vector>

#include "../../include/moni_hip.h"

struct PeBigPair {
    uint64_t pair = 0;                 // in: index of the pair in the resident batch (reads 2 * pair, 2 * pair + 1)
    // out: what pe_rec_t carries, with the CIGARs and alternative hits by value
    uint32_t status = 0, strand = 0;
    int32_t tot = 0, score2 = 0, score2_m[2] = {0, 0}, sub_n = 0;
    long long dist = 0;
    int32_t mate_score[2] = {0, 0};
    uint32_t filled[2] = {0, 0}, orphan[2] = {0, 0};
    uint64_t ref_pos[2] = {0, 0};
    int32_t as[2] = {0, 0};
    std::vector<uint32_t> cig[2];
    std::vector<uint64_t> alt_pos[2];
    std::vector<int32_t> alt_score[2];
    uint64_t csv_filter = 0, csv_skipped = 0;      // `-c`: occurrences of the MEMs the direction / frequency filters dropped, chains check_paired_left_MEM skipped
};
// one batch of DP problems (operands: the resident reads and the index text): results + CIGAR pool
typedef std::function<int(const std::vector<moni_dp_task_t>&, std::vector<moni_dp_result_t>&, std::vector<uint32_t>&)> PeBigDp;

// pe_params: a pe_params_t (pe_core.h) whose index pointers are HOST pointers (pdir may be null); mems / rmo / aux / occs: the seeds of the
// batch on the host; offs: read offsets relative to the batch.  Returns 0 or a MONI_E* code; a pair beyond even these capacities keeps status 2.
int pe_big_run(const void* pe_params, size_t pe_params_size, const moni_mem_t* mems, const uint64_t* rmo, const uint32_t* aux, const uint64_t* occs,
               const uint64_t* offs, std::vector<PeBigPair>& pairs, const PeBigDp& dp);
