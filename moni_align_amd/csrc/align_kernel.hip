// align_kernel: the whole per-read path after seeding in ONE launch, one wavefront per read (persistent waves, reads taken
// in a grid-stride loop).  Lane 0 runs the read's state machine (align_core.h: frequency filter, chaining with the
// libstdc++ sort emulation, chain-selection loop, fill_chain, CIGAR stitching); whenever it needs DP results all 64 lanes
// run those problems (extz_wave), operands taken in place from the resident reads and the index text.  Seeds are read
// where the seeding kernels left them in HBM: nothing goes to the host between stages.  Output per read: a fixed record
// plus CIGAR / alternative-hit entries in bump-allocated pools; MD/NM, MAPQ and SAM text are host work.
// A read that does not fit the kernel's capacities is marked (status 2) and taken by the host pipeline (align_host.hpp).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "align_core.h"

#define AK_CIG_CAP 1024u                     // CIGAR slots per DP problem of a round (qlen + tlen + 2 <= this)
#define AK_DIRS_CAP (384u * 1024u)           // direction bytes of one CIGAR problem

struct moni_aln_rec_t {                      // one per read
    uint32_t status;                         // 0 not aligned, 1 aligned, 2 overflow (host pipeline)
    uint32_t strand;
    uint64_t ref_pos;
    int32_t score, score2;
    uint32_t n_cigar, n_alt;
    uint64_t cigar_off, alt_off;             // into the pools
};
struct moni_alt_t { uint64_t pos; int32_t score; int32_t pad; };

struct ak_scratch_t {                        // per persistent wave, in HBM
    ac_ws_t ws;
    uint32_t cig[AC_MAX_TASKS * AK_CIG_CAP];
    uint8_t dirs[AK_DIRS_CAP];
};

struct ak_args_t {
    ac_params_t P;
    dp_launch_t D;                           // scoring + reads/text pointers (order/tasks/results unused)
    const moni_mem_t* mems;
    const uint64_t* occs;
    const uint64_t* read_mem_off;
    const uint64_t* offs;
    const int32_t* min_score_of_len;         // 20 + 8*log(l), computed on the host (libm) per read length
    uint32_t max_len;
    uint64_t read_lo, n_reads;               // this launch takes reads [read_lo, read_lo + n_reads) of the resident batch
    ak_scratch_t* scratch;
    moni_aln_rec_t* recs;
    uint32_t* cig_pool; uint64_t cig_cap;
    moni_alt_t* alt_pool; uint64_t alt_cap;
    unsigned long long* cursors;             // [0] cigar pool, [1] alt pool, [2] DP problems, [3] DP cells, [4] next read, [5..7] cycles: init, drive, dp
};

extern "C" __global__ void __launch_bounds__(64)
align_kernel(const ak_args_t A) {
    __shared__ dp_lds_t L;
    __shared__ moni_dp_task_t s_tasks[AC_MAX_TASKS];
    __shared__ moni_dp_result_t s_res[AC_MAX_TASKS];
    __shared__ uint32_t s_n, s_go;
    const int lane = threadIdx.x;
    ak_scratch_t* __restrict__ S = A.scratch + blockIdx.x;
    ac_ws_t& W = S->ws;
    unsigned long long n_dp = 0, n_cells = 0, cy_init = 0, cy_drive = 0, cy_dp = 0;
    __shared__ unsigned long long s_read;
    while (true) {
        // dynamic read queue: reads differ a lot in the number of chains they score
        if (lane == 0) s_read = atomicAdd(&A.cursors[4], 1ull);
        __syncthreads();
        if (s_read >= A.n_reads) break;
        const uint64_t r = A.read_lo + s_read;
        if (lane == 0) {
            const long long c0 = clock64();
            W.off = A.offs[r]; W.m = (uint32_t)(A.offs[r + 1] - A.offs[r]);
            W.min_score = A.min_score_of_len[W.m <= A.max_len ? W.m : A.max_len];
            s_n = 0; s_go = 0;
            const bool chained = ac_init(W, A.P, A.mems, A.read_mem_off[r], A.read_mem_off[r + 1], A.occs);
            const long long c1 = clock64();
            cy_init += (unsigned long long)(c1 - c0);
            if (chained) {
                ac_drive(W, A.P, nullptr, nullptr);
                cy_drive += (unsigned long long)(clock64() - c1);
                if (!W.overflow && W.stage != AC_DONE) {
                    s_n = W.n_tasks; s_go = 1;
                    for (uint32_t t = 0; t < W.n_tasks; ++t) s_tasks[t] = W.tasks[t];
                }
            }
        }
        __syncthreads();
        while (s_go) {
            const uint32_t nt = s_n;
            bool too_big = false;
            const long long d0 = clock64();
            for (uint32_t t = 0; t < nt; ++t) {
                const moni_dp_task_t task = s_tasks[t];
                const bool with_cigar = !(task.flag & DP_EZ_SCORE_ONLY);
                if (task.qlen > DP_MAX_QLEN || task.tlen > 512 || (uint32_t)(task.qlen + task.tlen + 2) > AK_CIG_CAP ||
                    (with_cigar && (uint64_t)(task.qlen + task.tlen - 1) * (uint64_t)task.tlen > AK_DIRS_CAP)) { too_big = true; break; }
                moni_dp_result_t R;
                uint32_t* cg = S->cig + (size_t)t * AK_CIG_CAP;
                extz_wave_lds(A.D, task, L, S->dirs, cg, R);
                R.cigar_off = t * AK_CIG_CAP;
                if (lane == 0) { s_res[t] = R; ++n_dp; n_cells += (unsigned long long)(task.qlen > 0 ? task.qlen : 0) * (unsigned long long)(task.tlen > 0 ? task.tlen : 0); }
            }
            __syncthreads();
            if (lane == 0) {
                const long long d1 = clock64();
                cy_dp += (unsigned long long)(d1 - d0);
                if (too_big) W.overflow = 1;
                else ac_drive(W, A.P, s_res, S->cig);
                cy_drive += (unsigned long long)(clock64() - d1);
                s_go = (!W.overflow && W.stage != AC_DONE) ? 1u : 0u;
                s_n = W.n_tasks;
                if (s_go) for (uint32_t t = 0; t < W.n_tasks; ++t) s_tasks[t] = W.tasks[t];
            }
            __syncthreads();
        }
        if (lane == 0) {
            moni_aln_rec_t rec;
            rec.status = W.overflow ? 2u : (W.aligned ? 1u : 0u);
            rec.strand = W.fill.strand; rec.ref_pos = W.fill.ref_pos; rec.score = W.fill.score; rec.score2 = W.score2;
            rec.n_cigar = 0; rec.n_alt = 0; rec.cigar_off = 0; rec.alt_off = 0;
            if (rec.status == 1) {
                const unsigned long long co = atomicAdd(&A.cursors[0], (unsigned long long)W.n_cigar);
                const unsigned long long ao = atomicAdd(&A.cursors[1], (unsigned long long)W.n_alt);
                if (co + W.n_cigar > A.cig_cap || ao + W.n_alt > A.alt_cap) rec.status = 2;      // pool too small: let the host pipeline redo the read
                else {
                    rec.n_cigar = W.n_cigar; rec.cigar_off = co; rec.n_alt = W.n_alt; rec.alt_off = ao;
                    for (uint32_t k = 0; k < W.n_cigar; ++k) A.cig_pool[co + k] = W.cigar[k];
                    for (uint32_t k = 0; k < W.n_alt; ++k) { moni_alt_t x; x.pos = W.alt_pos[k]; x.score = W.alt_score[k]; x.pad = 0; A.alt_pool[ao + k] = x; }
                }
            }
            A.recs[r - A.read_lo] = rec;
        }
        __syncthreads();
    }
    if (lane == 0) { atomicAdd(&A.cursors[2], n_dp); atomicAdd(&A.cursors[3], n_cells); atomicAdd(&A.cursors[5], cy_init); atomicAdd(&A.cursors[6], cy_drive); atomicAdd(&A.cursors[7], cy_dp); }
}
