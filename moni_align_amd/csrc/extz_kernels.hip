// extz_kernel: ksw_extz2_sse (lh3/ksw2, absent submodule; semantics per SURVEY.md App. A and the call sites
// include/aligner/aligner_ksw2.hpp:2812,2844,2965,2988,3015) as a hand-written anti-diagonal DP for gfx950.
//
// One 64-lane wavefront (= one workgroup) per DP problem.  Target rows live on lanes: lane l owns rows
// l, l+64, ... (NCH chunks), so H(i,j-1) and F(i,j-1) are the lane's own registers from the previous
// anti-diagonal and H(i-1,j), H(i-1,j-1), E(i-1,j) come from lane l-1 by a one-lane shuffle (DPP), with a
// readlane from lane 63 of the previous chunk at chunk seams.  No LDS traffic in the recurrence except the
// query byte; exact int32 arithmetic (the SSE code's 8-bit difference form is the same recurrence).
// Direction bytes (ksw2 encoding: bits 0-2 source, 0x08 E continues, 0x10 F continues) go to a per-task
// global scratch row-major by anti-diagonal, and lane 0 backtracks them exactly like ksw_backtrack.
// Integer DP: VALU-bound, reported in GCUPS; no MFMA.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "../../include/moni_hip.h"

#define DP_NEG_INF (-0x40000000)
#define DP_EZ_SCORE_ONLY 0x01
#define DP_EZ_RIGHT 0x02
#define DP_EZ_EXTZ_ONLY 0x40
#define DP_MAX_QLEN 2048
// moni_dp_task_t::reserved carries the operand mode (0 = nt4 codes in qseq/tseq, ascending: the plain ksw2 call)
#define DP_Q_READS 0x01   // query bytes come from the resident read batch (ASCII -> nt4, aligner_ksw2.hpp:3273 table)
#define DP_Q_REV   0x02   // query[k] = src[q_off - k]
#define DP_Q_COMP  0x04   // complement (the reverse-complement strand: kpbseq.h:120-137 then nt4)
#define DP_T_TEXT  0x08   // target bytes come from the index text (ASCII -> nt4)
#define DP_T_REV   0x10   // target[k] = src[t_off - k]   (left context, aligner_ksw2.hpp:2803-2805)

struct dp_launch_t {
    const uint8_t* qseq;
    const uint8_t* tseq;
    const moni_dp_task_t* tasks;
    const uint32_t* order;        // task indices of this launch
    const uint64_t* dir_off;      // per task: offset of its direction bytes (CIGAR tasks)
    const uint64_t* cig_off;      // per task: offset of its temporary CIGAR slots
    uint8_t* dirs;
    uint32_t* cig_tmp;
    moni_dp_result_t* results;
    int32_t sc_mch, sc_mis, sc_N, wild;
    int32_t qo, e, end_bonus;
    const uint8_t* reads;         // resident read batch (DP_Q_READS)
    const uint8_t* text;          // index text (DP_T_TEXT)
    uint64_t n_text;
    uint64_t reads_limit, text_limit;   // bytes of the two buffers that may be read as aligned 8-byte words (their padding included)
};

__device__ __forceinline__ uint32_t dp_nt4(uint32_t b) {      // seq_nt4_table, aligner_ksw2.hpp:3272-3288
    if (b < 4) return b;
    const uint32_t u = b & 0xDFu;                              // fold case
    return u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : 4u;
}

__device__ __forceinline__ int32_t dp_bound(int32_t k, int32_t qo, int32_t e) {   // H(k,-1) = H(-1,k), k >= -1
    return k < 0 ? 0 : -(qo + (k + 1) * e);
}

__device__ __forceinline__ long long wave_max_i64(long long v) {
    for (int o = 32; o > 0; o >>= 1) {
        const long long w = __shfl_xor(v, o);
        v = w > v ? w : v;
    }
    return v;
}

// One DP problem on one wavefront.  qs: LDS scratch for the query codes (>= qlen bytes); dir: direction bytes of this
// problem ((qlen+tlen-1)*tlen, CIGAR problems only); cg: CIGAR slots (qlen+tlen+2).  Every lane returns the same R.
template <int NCH>
__device__ __forceinline__ void extz_wave(const dp_launch_t& P, const moni_dp_task_t task, uint8_t* __restrict__ qs,
                                          uint8_t* __restrict__ dir_base, uint32_t* __restrict__ cg_base, moni_dp_result_t& R) {
    const int lane = threadIdx.x & 63;
    const int qlen = task.qlen, tlen = task.tlen, flag = task.flag;
    R.max = 0; R.max_q = R.max_t = R.mqe_t = R.mte_q = -1; R.mqe = R.mte = R.score = DP_NEG_INF;
    R.reach_end = 0; R.zdropped = 0; R.n_cigar = 0; R.cigar_off = 0;
    if (qlen <= 0 || tlen <= 0) return;                 // ksw_extz2_sse returns right after ksw_reset_extz
    const bool with_cigar = !(flag & DP_EZ_SCORE_ONLY);
    const bool right = (flag & DP_EZ_RIGHT) != 0;
    const int mode = task.reserved;
    if (mode & DP_Q_READS) {
        const uint8_t* __restrict__ q = P.reads + task.q_off;
        for (int k = lane; k < qlen; k += 64) {
            uint32_t c = dp_nt4((mode & DP_Q_REV) ? q[-(long)k] : q[k]);
            if ((mode & DP_Q_COMP) && c < 4) c = 3 - c;
            qs[k] = (uint8_t)c;
        }
    } else {
        const uint8_t* __restrict__ q = P.qseq + task.q_off;
        for (int k = lane; k < qlen; k += 64) qs[k] = (mode & DP_Q_REV) ? q[-(long)k] : q[k];
    }
    __syncthreads();
    const int32_t qo = P.qo, e = P.e;
    uint8_t* __restrict__ dir = with_cigar ? dir_base : nullptr;

    int32_t H1[NCH], H2[NCH], E1[NCH], F1[NCH];
    int32_t tcode[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int i = lane + 64 * k;
        H1[k] = dp_bound(i, qo, e);                     // H(i,-1)
        H2[k] = 0;
        E1[k] = DP_NEG_INF;
        F1[k] = DP_NEG_INF;
        int32_t tc = 255;
        if (i < tlen) {
            if (mode & DP_T_TEXT) {
                const uint64_t a = (mode & DP_T_REV) ? task.t_off - (uint64_t)i : task.t_off + (uint64_t)i;
                tc = (int32_t)dp_nt4(a < P.n_text ? P.text[a] : 0u);
            } else {
                tc = (int32_t)P.tseq[(mode & DP_T_REV) ? task.t_off - (uint64_t)i : task.t_off + (uint64_t)i];
            }
        }
        tcode[k] = tc;
    }
    // per-lane running results
    int32_t mqe_h = DP_NEG_INF, mqe_i = -1;            // best H(i, qlen-1) over own rows (rows ascend with the chunk index)
    long long max_key = -1;                             // (H, earliest diagonal, ksw2 lane order) of the best H > 0
    int32_t mte_h = DP_NEG_INF, mte_q = -1;
    const int en_r = (tlen - 1 + 16) / 16 * 16 - 1;     // ksw2 reports mte_q relative to the 16-padded band end

    const int n_diag = qlen + tlen - 1;
    for (int r = 0; r < n_diag; ++r) {
        const int st0 = r - qlen + 1 > 0 ? r - qlen + 1 : 0;
        const int en0 = r < tlen - 1 ? r : tlen - 1;
        const int en1 = st0 + (en0 - st0) / 4 * 4;
        // neighbour values from the previous anti-diagonal, for every chunk, before anything is updated
        int32_t uH1[NCH], uH2[NCH], uE[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            uH1[k] = __shfl_up(H1[k], 1);
            uH2[k] = __shfl_up(H2[k], 1);
            uE[k] = __shfl_up(E1[k], 1);
        }
#pragma unroll
        for (int k = NCH - 1; k >= 0; --k) {
            int32_t bH1, bH2, bE;
            if (k == 0) { bH1 = dp_bound(r, qo, e); bH2 = dp_bound(r - 1, qo, e); bE = DP_NEG_INF; }   // row -1: H(-1,j), H(-1,j-1)
            else { bH1 = __shfl(H1[k - 1], 63); bH2 = __shfl(H2[k - 1], 63); bE = __shfl(E1[k - 1], 63); }
            if (lane == 0) { uH1[k] = bH1; uH2[k] = bH2; uE[k] = bE; }
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (64 * k > en0 || 64 * k + 63 < st0) continue;          // wave-uniform: chunk outside this diagonal
            const int i = lane + 64 * k;
            const int j = r - i;
            if (i >= st0 && i <= en0) {
                const int32_t h_left = H1[k];
                const int32_t Eo = uH1[k] - qo;
                const int32_t E = (Eo > uE[k] ? Eo : uE[k]) - e;
                const int32_t Fo = h_left - qo;
                const int32_t F = (Fo > F1[k] ? Fo : F1[k]) - e;
                const int32_t qc = qs[j];
                const int32_t tc = tcode[k];
                const int32_t s = (tc == P.wild || qc == P.wild) ? P.sc_N : (tc == qc ? P.sc_mch : P.sc_mis);
                int32_t z = uH2[k] + s;
                uint32_t d;
                if (!right) {
                    d = E > z ? 1u : 0u;
                    z = z > E ? z : E;
                    d = F > z ? 2u : d;
                    z = z > F ? z : F;
                    if (E > z - qo) d |= 0x08u;
                    if (F > z - qo) d |= 0x10u;
                } else {
                    d = z > E ? 0u : 1u;
                    z = z > E ? z : E;
                    d = z > F ? d : 2u;
                    z = z > F ? z : F;
                    if (E >= z - qo) d |= 0x08u;
                    if (F >= z - qo) d |= 0x10u;
                }
                H2[k] = h_left; H1[k] = z; E1[k] = E; F1[k] = F;
                if (with_cigar) dir[(size_t)r * tlen + i] = (uint8_t)d;
                if (j == qlen - 1 && z > mqe_h) { mqe_h = z; mqe_i = i; }
                if (i == tlen - 1 && z > mte_h) { mte_h = z; mte_q = r - en_r; }
                if (z > 0) {
                    int rank;
                    if (i == en0) rank = 0;
                    else if (i < en1) rank = 1 + ((i - st0) & 3) * 4096 + ((i - st0) >> 2);
                    else rank = 1 + 4 * 4096 + (i - en1);
                    if (r == 0) rank = 0;
                    const long long key = (long long)(((unsigned long long)(uint32_t)z << 32) | ((unsigned long long)(0xFFFF - r) << 16) |
                                                      (unsigned long long)(0xFFFF - rank));
                    if (key > max_key) max_key = key;
                }
            }
        }
    }
    // ---- wave reductions ----
    const long long mqe_key = wave_max_i64((long long)(((unsigned long long)(long long)mqe_h << 32) |
                                                       (unsigned long long)(uint32_t)(0x7FFFFFFF - (mqe_i < 0 ? 0x7FFFFFFF : mqe_i))));
    R.mqe = (int32_t)(mqe_key >> 32);
    R.mqe_t = 0x7FFFFFFF - (int32_t)(mqe_key & 0xFFFFFFFFll);
    max_key = wave_max_i64(max_key);
    if (max_key >= 0) {
        R.max = (int32_t)(max_key >> 32);
        const int rr = 0xFFFF - (int)((max_key >> 16) & 0xFFFF);
        const int rank = 0xFFFF - (int)(max_key & 0xFFFF);
        const int st0 = rr - qlen + 1 > 0 ? rr - qlen + 1 : 0;
        const int en0 = rr < tlen - 1 ? rr : tlen - 1;
        const int en1 = st0 + (en0 - st0) / 4 * 4;
        int t;
        if (rank == 0) t = rr == 0 ? 0 : en0;
        else if (rank < 1 + 4 * 4096) t = st0 + ((rank - 1) / 4096) + 4 * ((rank - 1) % 4096);
        else t = en1 + (rank - 1 - 4 * 4096);
        R.max_t = t; R.max_q = rr - t;
    }
    {   // the lane that owns row tlen-1 holds mte and the global score
        const int owner = (tlen - 1) & 63;
        int32_t sc = DP_NEG_INF;
#pragma unroll
        for (int k = 0; k < NCH; ++k) if ((tlen - 1) >> 6 == k) sc = H1[k];
        R.mte = __shfl(mte_h, owner);
        R.mte_q = __shfl(mte_q, owner);
        R.score = __shfl(sc, owner);
    }
    // ---- backtrack (ksw_backtrack, is_rot = 1) by lane 0 ----
    if (with_cigar) {
        __syncthreads();                                // direction bytes written by all lanes -> visible to lane 0
        int i0 = -1, j0 = -1;
        if (!(flag & DP_EZ_EXTZ_ONLY)) { i0 = tlen - 1; j0 = qlen - 1; }
        else if (R.mqe + P.end_bonus > R.max) { R.reach_end = 1; i0 = R.mqe_t; j0 = qlen - 1; }
        else if (R.max_t >= 0 && R.max_q >= 0) { i0 = R.max_t; j0 = R.max_q; }
        if (lane == 0 && i0 >= 0 && j0 >= 0) {
            uint32_t* __restrict__ cg = cg_base;
            int n = 0, i = i0, j = j0, state = 0;
            auto push = [&](uint32_t op, int len) {
                if (n == 0 || op != (cg[n - 1] & 0xf)) cg[n++] = (uint32_t)len << 4 | op;
                else cg[n - 1] += (uint32_t)len << 4;
            };
            while (i >= 0 && j >= 0) {
                const uint32_t tmp = dir[(size_t)(i + j) * tlen + i];
                if (state == 0) state = tmp & 7;
                else if (!(tmp >> (state + 2) & 1)) state = 0;
                if (state == 0) state = tmp & 7;
                if (state == 0) { push(0, 1); --i; --j; }
                else if (state == 1 || state == 3) { push(2, 1); --i; }
                else { push(1, 1); --j; }
            }
            if (i >= 0) push(2, i + 1);
            if (j >= 0) push(1, j + 1);
            for (int a = 0; a < n >> 1; ++a) { const uint32_t t2 = cg[a]; cg[a] = cg[n - 1 - a]; cg[n - 1 - a] = t2; }
            R.n_cigar = (uint32_t)n;
        }
    }
    R.n_cigar = (uint32_t)__shfl((int)R.n_cigar, 0);
    __syncthreads();                                    // qs / dir are reused by the next problem of this wave
}

template <int NCH>
__global__ void __launch_bounds__(64)
extz_kernel(const dp_launch_t P) {
    __shared__ uint8_t qs[DP_MAX_QLEN];
    const uint32_t tix = P.order[blockIdx.x];
    const moni_dp_task_t task = P.tasks[tix];
    const bool with_cigar = !(task.flag & DP_EZ_SCORE_ONLY);
    moni_dp_result_t R;
    extz_wave<NCH>(P, task, qs, with_cigar ? P.dirs + P.dir_off[tix] : nullptr, with_cigar ? P.cig_tmp + P.cig_off[tix] : nullptr, R);
    if (threadIdx.x == 0) P.results[tix] = R;
}

// out-of-line instances for callers that run many problem sizes from one kernel (align_kernel)
template <int NCH>
__device__ __attribute__((noinline)) void extz_wave_call(const dp_launch_t& P, const moni_dp_task_t task, uint8_t* qs, uint8_t* dir_base,
                                                         uint32_t* cg_base, moni_dp_result_t& R) {
    extz_wave<NCH>(P, task, qs, dir_base, cg_base, R);
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-tiled form of the same DP: H (two anti-diagonals), E, F and the target codes live in LDS, one entry per target row,
// and a diagonal is swept in 64-row chunks from the bottom up, so that a chunk reads its upper neighbour's previous-
// diagonal values before the chunk above overwrites them.  No register blocking over the target length: one code path
// for every problem size, ~40 VGPRs, so a kernel that runs DP problems of many sizes (align_kernel) keeps its occupancy.
// Results are bit-identical to extz_wave (same recurrence, same tie rules, same traceback).
// ------------------------------------------------------------------------------------------------------------------
#define DP_LDS_T 512
#define DP_LDS_Q 512                         // longest query of the LDS-tiled form
struct dp_lds_t {
    int32_t H[2][DP_LDS_T];
    int32_t E[DP_LDS_T];
    int32_t F[DP_LDS_T];
    uint8_t tc[DP_LDS_T];
    uint8_t qs[DP_LDS_Q];
    static constexpr bool in_lds = true;
};

// ST: where the per-row state lives - dp_lds_t (LDS; align_kernel, extz_lds_kernel) or dp_big_t (HBM; problems too large for it).
// tile: 2 KB of LDS for the traceback (may overlay ST's H buffers when those are in LDS: they are dead by then).
// TRACK: what is kept about the maximum cell (ez->max, max_t, max_q).  2 = everything, as ksw2 does (the stand-alone kernels);
// 1 = its value only - enough to decide reach_end, the function returns true when the traceback would have to start at the
// maximum cell and the caller runs it again with TRACK = 2; 0 = nothing (score-only problems of align_kernel, whose logic
// reads mqe / mqe_t / score only: the max fields of the result are then 0 / -1).
struct dp_brief_t { int32_t mqe, mqe_t, score; };      // what the aligner's logic reads of a score-only result

template <class ST, int TRACK>
__device__ __forceinline__ bool extz_wave_tiled(const dp_launch_t& P, const moni_dp_task_t task, ST& L, uint8_t* __restrict__ tile,
                                                uint8_t* __restrict__ dir_base, uint32_t* __restrict__ cg_base, moni_dp_result_t* __restrict__ out,
                                                uint32_t cigar_off = 0, dp_brief_t* brief = nullptr) {
    const int lane = threadIdx.x & 63;
    const int qlen = task.qlen, tlen = task.tlen, flag = task.flag;
    moni_dp_result_t R;                      // lane 0 writes it to *out (LDS or global) at the end
    R.max = 0; R.max_q = R.max_t = R.mqe_t = R.mte_q = -1; R.mqe = R.mte = R.score = DP_NEG_INF;
    R.reach_end = 0; R.zdropped = 0; R.n_cigar = 0; R.cigar_off = cigar_off;
    if (brief) { brief->mqe = R.mqe; brief->mqe_t = R.mqe_t; brief->score = R.score; }
    if (qlen <= 0 || tlen <= 0) { if (lane == 0) *out = R; return false; }
    const bool with_cigar = !(flag & DP_EZ_SCORE_ONLY);
    const bool right = (flag & DP_EZ_RIGHT) != 0;
    const int mode = task.reserved;
    const int32_t qo = P.qo, e = P.e;
    if (mode & DP_Q_READS) {
        const uint8_t* __restrict__ q = P.reads + task.q_off;
        for (int k = lane; k < qlen; k += 64) {
            uint32_t c = dp_nt4((mode & DP_Q_REV) ? q[-(long)k] : q[k]);
            if ((mode & DP_Q_COMP) && c < 4) c = 3 - c;
            L.qs[k] = (uint8_t)c;
        }
    } else {
        const uint8_t* __restrict__ q = P.qseq + task.q_off;
        for (int k = lane; k < qlen; k += 64) L.qs[k] = (mode & DP_Q_REV) ? q[-(long)k] : q[k];
    }
    for (int i = lane; i < tlen; i += 64) {
        uint32_t tc;
        if (mode & DP_T_TEXT) {
            const uint64_t a = (mode & DP_T_REV) ? task.t_off - (uint64_t)i : task.t_off + (uint64_t)i;
            tc = dp_nt4(a < P.n_text ? P.text[a] : 0u);
        } else tc = P.tseq[(mode & DP_T_REV) ? task.t_off - (uint64_t)i : task.t_off + (uint64_t)i];
        L.tc[i] = (uint8_t)tc;
        const int32_t b = dp_bound(i, qo, e);                 // H(i,-1) in both diagonal buffers
        L.H[0][i] = b; L.H[1][i] = b; L.E[i] = DP_NEG_INF; L.F[i] = DP_NEG_INF;
    }
    __syncthreads();
    uint8_t* __restrict__ dir = with_cigar ? dir_base : nullptr;
    const int32_t wild = P.wild, scN = P.sc_N, scM = P.sc_mch, scX = P.sc_mis;     // P lives in memory: read once
    bool any_wild = false;                                    // most problems have no N on either side
    for (int k = lane; k < qlen; k += 64) any_wild |= L.qs[k] == wild;
    for (int i = lane; i < tlen; i += 64) any_wild |= L.tc[i] == wild;
    const bool has_wild = __ballot(any_wild) != 0ull;
    // running maximum per lane as (value, diagonal, row); the reference's in-diagonal visiting order only decides ties
    // inside one diagonal, resolved in the rare branch below and when the lanes are merged
    auto rank_of = [&](int rr, int i) {
        const int st0 = rr - qlen + 1 > 0 ? rr - qlen + 1 : 0;
        const int en0 = rr < tlen - 1 ? rr : tlen - 1;
        const int en1 = st0 + (en0 - st0) / 4 * 4;
        if (rr == 0 || i == en0) return 0;
        if (i < en1) return 1 + ((i - st0) & 3) * 4096 + ((i - st0) >> 2);
        return 1 + 4 * 4096 + (i - en1);
    };
    int32_t max_z = 0, max_r = 0, max_i = -1;
    int32_t mte_h = DP_NEG_INF, mte_q = -1, last_h = DP_NEG_INF;      // uniform: read back from the last row after each diagonal
    const int en_r = (tlen - 1 + 16) / 16 * 16 - 1;
    const int n_diag = qlen + tlen - 1;
    // one anti-diagonal: its cells st0..en0 are laid over the lanes from st0 upwards (the state lives in LDS, so the
    // row -> lane map may slide), 64 rows at a time from the bottom up.  PAR = r & 1 picks the H buffer at compile time.
    auto diag = [&](const int r, auto PAR) __attribute__((always_inline)) {
        constexpr int par = decltype(PAR)::value;
        int32_t* __restrict__ Hn = L.H[par];                 // holds diagonal r-2, receives diagonal r
        const int32_t* __restrict__ Hp = L.H[par ^ 1];       // diagonal r-1
        const int st0 = r - qlen + 1 > 0 ? r - qlen + 1 : 0;
        const int en0 = r < tlen - 1 ? r : tlen - 1;
        const int32_t b_r = dp_bound(r, qo, e), b_r1 = dp_bound(r - 1, qo, e);      // row -1: H(-1,r), H(-1,r-1)
        struct cell_in { int32_t h_left, f_old, uH1, uH2, uE, qc, tc; };
        auto load = [&](const int i) __attribute__((always_inline)) {      // i: a live row of this diagonal
            cell_in c;
            const int up = i > 0 ? i - 1 : 0;
            c.h_left = Hp[i];
            c.f_old = L.F[i];
            c.uH1 = Hp[up]; c.uH2 = Hn[up]; c.uE = L.E[up];                  // Hn[i-1] is still H(i-1,-1) when j == 0
            c.qc = L.qs[r - i];
            c.tc = L.tc[i];
            return c;
        };
        auto finish = [&](const int i, const bool act, const cell_in& c) __attribute__((always_inline)) {
            const int32_t uH1 = i > 0 ? c.uH1 : b_r, uH2 = i > 0 ? c.uH2 : b_r1, uE = i > 0 ? c.uE : DP_NEG_INF;
            const int32_t Eo = uH1 - qo;
            const int32_t E = (Eo > uE ? Eo : uE) - e;
            const int32_t Fo = c.h_left - qo;
            const int32_t F = (Fo > c.f_old ? Fo : c.f_old) - e;
            int32_t s = c.tc == c.qc ? scM : scX;
            if (has_wild) s = (c.tc == wild || c.qc == wild) ? scN : s;
            int32_t z = uH2 + s;
            uint32_t d = 0;
            if (with_cigar) {
                if (!right) {
                    d = E > z ? 1u : 0u; z = z > E ? z : E; d = F > z ? 2u : d; z = z > F ? z : F;
                    d |= (E > z - qo) ? 0x08u : 0u;
                    d |= (F > z - qo) ? 0x10u : 0u;
                } else {
                    d = z > E ? 0u : 1u; z = z > E ? z : E; d = z > F ? d : 2u; z = z > F ? z : F;
                    d |= (E >= z - qo) ? 0x08u : 0u;
                    d |= (F >= z - qo) ? 0x10u : 0u;
                }
            } else { z = z > E ? z : E; z = z > F ? z : F; }
            if (act) {
                if (with_cigar) dir[(uint32_t)r * (uint32_t)tlen + (uint32_t)i] = (uint8_t)d;
                Hn[i] = z; L.E[i] = E; L.F[i] = F;
                if (TRACK == 2) {
                    if (z > max_z) { max_z = z; max_r = r; max_i = i; }
                    else if (z == max_z && max_r == r && z > 0 && rank_of(r, i) < rank_of(r, max_i)) max_i = i;
                } else if (TRACK == 1) max_z = z > max_z ? z : max_z;
            }
        };
        int c = (en0 - st0) >> 6;
        for (; c >= 1; c -= 2) {            // two 64-row chunks at a time: both read before either writes, their chains overlap
            const int i_hi = st0 + 64 * c + lane, i_lo = i_hi - 64;
            const bool act_hi = i_hi <= en0;                                  // the lower chunk is full
            const cell_in in_hi = load(act_hi ? i_hi : en0);
            const cell_in in_lo = load(i_lo);
            finish(i_hi, act_hi, in_hi);
            finish(i_lo, true, in_lo);
        }
        if (c == 0) {
            const int i = st0 + lane;
            const bool act = i <= en0;
            const cell_in in = load(act ? i : en0);
            finish(i, act, in);
        }
        if (!ST::in_lds) __threadfence_block();
        if (TRACK == 2 ? en0 == tlen - 1 : r == n_diag - 1) {    // the last row's cell of this diagonal (mte; score is its last value)
            const int32_t z = Hn[tlen - 1];
            last_h = z;
            if (z > mte_h) { mte_h = z; mte_q = r - en_r; }
        }
    };
    {
        int r = 0;
        for (; r + 1 < n_diag; r += 2) { diag(r, std::integral_constant<int, 0>{}); diag(r + 1, std::integral_constant<int, 1>{}); }
        if (r < n_diag) diag(r, std::integral_constant<int, 0>{});
    }
    __syncthreads();
    // mqe: H(i, qlen-1) of every row is still in the buffer of the diagonal it was written on
    int32_t mqe_h = DP_NEG_INF, mqe_i = -1;
    for (int i = lane; i < tlen; i += 64) {
        const int32_t h = L.H[(i + qlen - 1) & 1][i];
        if (h > mqe_h) { mqe_h = h; mqe_i = i; }
    }
    long long max_key = -1;
    if (max_i >= 0) max_key = (long long)(((unsigned long long)(uint32_t)max_z << 32) | ((unsigned long long)(0xFFFF - max_r) << 16) |
                                          (unsigned long long)(0xFFFF - rank_of(max_r, max_i)));
    const long long mqe_key = wave_max_i64((long long)(((unsigned long long)(long long)mqe_h << 32) |
                                                       (unsigned long long)(uint32_t)(0x7FFFFFFF - (mqe_i < 0 ? 0x7FFFFFFF : mqe_i))));
    R.mqe = (int32_t)(mqe_key >> 32);
    R.mqe_t = 0x7FFFFFFF - (int32_t)(mqe_key & 0xFFFFFFFFll);
    if (TRACK == 1) {
        int32_t v = max_z;
        for (int o = 32; o > 0; o >>= 1) { const int32_t w2 = __shfl_xor(v, o); v = w2 > v ? w2 : v; }
        R.max = v;
    }
    if (TRACK == 2) max_key = wave_max_i64(max_key);
    if (TRACK == 2 && max_key >= 0) {
        R.max = (int32_t)(max_key >> 32);
        const int rr = 0xFFFF - (int)((max_key >> 16) & 0xFFFF);
        const int rank = 0xFFFF - (int)(max_key & 0xFFFF);
        const int st0 = rr - qlen + 1 > 0 ? rr - qlen + 1 : 0;
        const int en0 = rr < tlen - 1 ? rr : tlen - 1;
        const int en1 = st0 + (en0 - st0) / 4 * 4;
        int t;
        if (rank == 0) t = rr == 0 ? 0 : en0;
        else if (rank < 1 + 4 * 4096) t = st0 + ((rank - 1) / 4096) + 4 * ((rank - 1) % 4096);
        else t = en1 + (rank - 1 - 4 * 4096);
        R.max_t = t; R.max_q = rr - t;
    }
    R.mte = mte_h; R.mte_q = mte_q; R.score = last_h;
    if (with_cigar) {
        __syncthreads();
        int i0 = -1, j0 = -1;
        if (!(flag & DP_EZ_EXTZ_ONLY)) { i0 = tlen - 1; j0 = qlen - 1; }
        else if (R.mqe + P.end_bonus > R.max) { R.reach_end = 1; i0 = R.mqe_t; j0 = qlen - 1; }
        else if (TRACK != 2) return true;                   // the traceback starts at the maximum cell: its position is needed
        else if (R.max_t >= 0 && R.max_q >= 0) { i0 = R.max_t; j0 = R.max_q; }
        if (i0 >= 0 && j0 >= 0) {
            // traceback, the whole wave in step (i, j and the state are uniform): direction bytes come through LDS in tiles of
            // 64 anti-diagonals x 32 rows ending at the current cell (one round trip to HBM per tile instead of one per step);
            // the CIGAR run being built stays in registers, lane 0 stores an entry when the operation changes
            uint32_t* __restrict__ cg = cg_base;
            int n = 0, i = i0, j = j0, state = 0;
            uint32_t cur_op = 0xFu, cur_len = 0;
            auto push = [&](uint32_t op, int len) {
                if (op == cur_op) cur_len += (uint32_t)len;
                else { if (cur_op != 0xFu) { if (lane == 0) cg[n] = cur_len << 4 | cur_op; ++n; } cur_op = op; cur_len = (uint32_t)len; }
            };
            while (i >= 0 && j >= 0) {
                const int r_top = i + j, i_lo = i > 31 ? i - 31 : 0;
                __syncthreads();
                _Pragma("unroll 8")
                for (int s2 = 0; s2 < 32; ++s2) {                   // lane: diagonal r_top - (2*s2 + lane/32), row i_lo + lane%32
                    const int d = 2 * s2 + (lane >> 5), rd = r_top - d, row = i_lo + (lane & 31);
                    uint8_t v = 0;
                    if (rd >= 0 && row <= i) v = dir[(uint32_t)rd * (uint32_t)tlen + (uint32_t)row];
                    tile[d * 32 + (lane & 31)] = v;
                }
                __syncthreads();
                while (i >= i_lo && j >= 0 && i + j > r_top - 64) {
                    const uint32_t tmp = tile[(r_top - (i + j)) * 32 + (i - i_lo)];
                    if (state == 0) state = tmp & 7;
                    else if (!(tmp >> (state + 2) & 1)) state = 0;
                    if (state == 0) state = tmp & 7;
                    if (state == 0) { push(0, 1); --i; --j; }
                    else if (state == 1 || state == 3) { push(2, 1); --i; }
                    else { push(1, 1); --j; }
                }
            }
            if (i >= 0) push(2, i + 1);
            if (j >= 0) push(1, j + 1);
            if (cur_op != 0xFu) { if (lane == 0) cg[n] = cur_len << 4 | cur_op; ++n; }
            __syncthreads();
            if (lane == 0) for (int a2 = 0; a2 < n >> 1; ++a2) { const uint32_t t2 = cg[a2]; cg[a2] = cg[n - 1 - a2]; cg[n - 1 - a2] = t2; }
            R.n_cigar = (uint32_t)n;
        }
    }
    if (brief) { brief->mqe = R.mqe; brief->mqe_t = R.mqe_t; brief->score = R.score; }      // uniform across the wave
    if (lane == 0) *out = R;
    __syncthreads();
    return false;
}

__device__ __attribute__((noinline)) void extz_wave_lds(const dp_launch_t& P, const moni_dp_task_t task, dp_lds_t& L,
                                                        uint8_t* __restrict__ dir_base, uint32_t* __restrict__ cg_base, moni_dp_result_t* __restrict__ out) {
    extz_wave_tiled<dp_lds_t, 2>(P, task, L, reinterpret_cast<uint8_t*>(&L.H[0][0]), dir_base, cg_base, out);
}
// align_kernel's form: no maximum-cell bookkeeping for score-only problems, its value only for traceback problems (and the full
// run again in the rare case the traceback has to start there)
__device__ __attribute__((noinline)) dp_brief_t extz_wave_lds_lite(const dp_launch_t& P, const moni_dp_task_t task, dp_lds_t& L,
                                                                   uint8_t* __restrict__ dir_base, uint32_t* __restrict__ cg_base, moni_dp_result_t* __restrict__ out,
                                                                   uint32_t cigar_off) {
    uint8_t* tile = reinterpret_cast<uint8_t*>(&L.H[0][0]);
    dp_brief_t b;
    if (task.flag & DP_EZ_SCORE_ONLY) { extz_wave_tiled<dp_lds_t, 0>(P, task, L, tile, dir_base, cg_base, out, cigar_off, &b); return b; }
    if (extz_wave_tiled<dp_lds_t, 1>(P, task, L, tile, dir_base, cg_base, out, cigar_off, &b)) {
        __syncthreads();
        extz_wave_tiled<dp_lds_t, 2>(P, task, L, tile, dir_base, cg_base, out, cigar_off, &b);
    }
    return b;
}

// problems beyond the LDS form's 512 target rows / 512 query bases (long reads in the host pipeline): same code, state in HBM
#define DP_BIG_T 4096
#define DP_BIG_Q 8192
struct dp_big_t {
    int32_t H[2][DP_BIG_T];
    int32_t E[DP_BIG_T];
    int32_t F[DP_BIG_T];
    uint8_t tc[DP_BIG_T];
    uint8_t qs[DP_BIG_Q];
    static constexpr bool in_lds = false;     // lanes hand rows to each other through HBM: drain the stores after every diagonal
};

__global__ void __launch_bounds__(64)
extz_big_kernel(const dp_launch_t P, dp_big_t* __restrict__ states) {
    __shared__ uint8_t tile[2048];
    const uint32_t tix = P.order[blockIdx.x];
    const moni_dp_task_t task = P.tasks[tix];
    const bool with_cigar = !(task.flag & DP_EZ_SCORE_ONLY);
    extz_wave_tiled<dp_big_t, 2>(P, task, states[blockIdx.x], tile, with_cigar ? P.dirs + P.dir_off[tix] : nullptr, with_cigar ? P.cig_tmp + P.cig_off[tix] : nullptr, &P.results[tix]);
}

__global__ void __launch_bounds__(64)
extz_lds_kernel(const dp_launch_t P) {
    __shared__ dp_lds_t L;
    const uint32_t tix = P.order[blockIdx.x];
    const moni_dp_task_t task = P.tasks[tix];
    const bool with_cigar = !(task.flag & DP_EZ_SCORE_ONLY);
    extz_wave_lds(P, task, L, with_cigar ? P.dirs + P.dir_off[tix] : nullptr, with_cigar ? P.cig_tmp + P.cig_off[tix] : nullptr, &P.results[tix]);
}
