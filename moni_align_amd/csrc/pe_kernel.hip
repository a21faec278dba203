// pe_align_kernel: the paired-end path after seeding, persistent waves, AK_NL pairs in flight per wavefront - the same shape as
// align_kernel (every lane runs one pair's state machine, pe_core.h, out of its own slot in HBM; whenever pairs need DP results the
// whole wave solves their problems one after the other, extz_wave_lds_lite).  Reads 2p and 2p + 1 of the resident batch are the mates
// of pair p; the seeds are read where the seeding kernels left them (the four (mate, strand) lists of aligner_ksw2.hpp:1012-1040 are
// put in the reference's order by pe_init).  Output per pair: a fixed record plus the two CIGARs and the alternative hits of each mate
// in bump-allocated pools; lift-over, MD/NM, MAPQ and the SAM text of a pair are host work (pe_host.hpp).
// finalize == 0 is the learn pass of learn_fragment_model: the record carries best_scores[0] / [1] only.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pe_core.h"

// klib's ksw_align(KSW_XSTART) for the orphan search (pe_core.h queues it as a DP_EZ_LOCAL request): ksw_i16's local alignment, rows =
// target, with its tie rules (the first row that reaches the maximum, the leftmost query position in it), then the same over the reversed
// prefixes up to the first row that reaches the score (tb / qb).  One lane per request, rolling rows in the lane's workspace in HBM: the
// requests are rare (pairs that chain but fail jointly) and each lane's pair waits for nothing else.
#define PE_SW_QMAX 4096
struct pe_sw_ws_t { int32_t H0[PE_SW_QMAX], H1[PE_SW_QMAX], E[PE_SW_QMAX], HM[PE_SW_QMAX]; uint8_t qc[PE_SW_QMAX]; };

// one pass; reverse: query[j] = q(q_last - j), target[i] = i <= t_last ? T(t_last - i) : T(i) (klib reverses the two prefixes in place)
__device__ __attribute__((noinline)) void pe_sw_pass(const dp_launch_t& D, pe_sw_ws_t& ws, const moni_dp_task_t& task, int qlen, int tlen, bool reverse, int q_last, int t_last,
                                                     int endsc, int& score, int& te_out, int& qe_out) {
    const int mode = task.reserved;
    for (int j = 0; j < qlen; ++j) {
        const int k = reverse ? q_last - j : j;
        uint32_t c = dp_nt4(D.reads[(mode & DP_Q_REV) ? task.q_off - (uint64_t)k : task.q_off + (uint64_t)k]);
        if ((mode & DP_Q_COMP) && c < 4) c = 3 - c;
        ws.qc[j] = (uint8_t)c; ws.H0[j] = 0; ws.H1[j] = 0; ws.E[j] = 0; ws.HM[j] = 0;
    }
    const int gape = D.e, gapoe = D.qo + D.e;
    int32_t* H0 = ws.H0; int32_t* H1 = ws.H1;
    int gmax = 0, te = -1;
    for (int i = 0; i < tlen; ++i) {
        const int ti = (reverse && i <= t_last) ? t_last - i : i;
        const uint64_t ta = task.t_off + (uint64_t)ti;
        const uint32_t tc = dp_nt4(ta < D.n_text ? D.text[ta] : 0u);
        int f = 0, imax = 0, diag = 0;
        for (int j = 0; j < qlen; ++j) {
            const uint32_t qc = ws.qc[j];
            const int sc = (tc >= 4 || qc >= 4) ? 0 : (tc == qc ? D.sc_mch : D.sc_mis);
            int h = diag + sc;
            diag = H0[j];
            const int e = ws.E[j];
            h = h > e ? h : e; h = h > f ? h : f;
            H1[j] = h;
            imax = imax > h ? imax : h;
            int hh = h - gapoe; hh = hh > 0 ? hh : 0;
            int e2 = e - gape; e2 = e2 > 0 ? e2 : 0;
            ws.E[j] = e2 > hh ? e2 : hh;
            f -= gape; f = f > 0 ? f : 0; f = f > hh ? f : hh;
        }
        if (imax > gmax) { gmax = imax; te = i; for (int j = 0; j < qlen; ++j) ws.HM[j] = H1[j]; if (gmax >= endsc) break; }
        int32_t* tmp = H0; H0 = H1; H1 = tmp;
    }
    score = gmax; te_out = te;
    int mx = -1, qe = -1;
    for (int j = 0; j < qlen; ++j) if (ws.HM[j] > mx) { mx = ws.HM[j]; qe = j; }
    qe_out = qe;
}
__device__ __attribute__((noinline)) void pe_sw_local(const dp_launch_t& D, pe_sw_ws_t& ws, const moni_dp_task_t& task, moni_dp_result_t* out) {
    moni_dp_result_t r;
    r.max = 0; r.max_q = r.max_t = -1; r.mqe = 0; r.mqe_t = -1; r.mte = -1; r.mte_q = -1; r.score = 0; r.reach_end = 0; r.zdropped = 0; r.n_cigar = 0; r.cigar_off = 0;
    if (task.qlen > 0 && task.qlen <= PE_SW_QMAX && task.tlen > 0) {
        int score, te, qe;
        pe_sw_pass(D, ws, task, task.qlen, task.tlen, false, 0, 0, 0x10000, score, te, qe);
        int s2, te2, qe2;
        pe_sw_pass(D, ws, task, qe + 1, task.tlen, true, qe, te, score, s2, te2, qe2);
        r.score = score; r.max_t = te; r.max_q = qe;
        if (score == s2) { r.mte = te - te2; r.mte_q = qe - qe2; }
    }
    *out = r;
}
// The same pass by the whole wave (the kernel's default): 128 target rows at a time, lane l on rows i0 + 2 l and i0 + 2 l + 1, the cells of an anti-diagonal in one
// step (lane l on column s - l); a lane takes H / E of the row above its first row from lane l - 1 by one shuffle per step (lane 0: from the previous tile's
// bottom row in LDS), keeps F and the diagonal to itself, and tracks its row's maximum and the leftmost column that reaches it; after a tile
// the rows are visited in order for klib's rule (the first row that exceeds every earlier one; stop at endsc).  ~(qlen + 63) * tlen / 64 steps
// instead of qlen * tlen cells by one lane.  qlen <= DP_LDS_Q; the dataflow was checked lane by lane against the oracle before it was written here.
__device__ __attribute__((noinline)) void pe_sw_pass_wave(const dp_launch_t& D, dp_lds_t& L, const moni_dp_task_t& task, int qlen, int tlen, bool reverse, int q_last, int t_last,
                                                          int endsc, int& score, int& te_out, int& qe_out) {
    const int lane = threadIdx.x & 63;
    const int mode = task.reserved;
    int32_t* const Hb = L.H[0]; int32_t* const Eb = L.E; uint8_t* const qc = L.qs;
    __syncthreads();
    for (int j = lane; j < qlen; j += 64) {
        const int k = reverse ? q_last - j : j;
        uint32_t c = dp_nt4(D.reads[(mode & DP_Q_REV) ? task.q_off - (uint64_t)k : task.q_off + (uint64_t)k]);
        if ((mode & DP_Q_COMP) && c < 4) c = 3 - c;
        qc[j] = (uint8_t)c; Hb[j] = 0; Eb[j] = 0;
    }
    __syncthreads();
    const int gape = D.e, gapoe = D.qo + D.e;
    int gmax = 0, te = -1, qe = -1;
    bool stop = false;
    // Two target rows per lane (a = i0 + 2 l, b = a + 1; 128 rows per tile): lane l is on column j = s - l of both in step s - row b's cell comes right after row a's,
    // whose H / E it takes from registers - and hands row b's H / E to lane l + 1 by one shuffle per step: half the tiles, the same qlen + 63 steps per tile.
    for (int i0 = 0; i0 < tlen && !stop; i0 += 128) {
        const int ia = i0 + 2 * lane, ib = ia + 1;
        const bool ok_a = ia < tlen, ok_b = ib < tlen;
        uint32_t tca = 4, tcb = 4;
        if (ok_a) { const int ti = (reverse && ia <= t_last) ? t_last - ia : ia; const uint64_t ta = task.t_off + (uint64_t)ti; tca = dp_nt4(ta < D.n_text ? D.text[ta] : 0u); }
        if (ok_b) { const int ti = (reverse && ib <= t_last) ? t_last - ib : ib; const uint64_t ta = task.t_off + (uint64_t)ti; tcb = dp_nt4(ta < D.n_text ? D.text[ta] : 0u); }
        int pubH = 0, pubE = 0, upH_prev = 0, ha_prev = 0, fa = 0, fb = 0, rmax_a = 0, rarg_a = -1, rmax_b = 0, rarg_b = -1;
        const int n_steps = qlen + 63;
        for (int s = 0; s < n_steps; ++s) {
            const int inH = __shfl_up(pubH, 1), inE = __shfl_up(pubE, 1);
            const int j = s - lane;
            const bool col_ok = j >= 0 && j < qlen;
            int upE, diag;
            if (lane == 0) { upE = col_ok ? Eb[j] : 0; diag = (j >= 1 && j < qlen) ? Hb[j - 1] : 0; }
            else { upE = inE; diag = j >= 1 ? upH_prev : 0; upH_prev = inH; }
            if (ok_a && col_ok) {
                if (j == 0) { fa = 0; fb = 0; }
                const uint32_t q = qc[j];
                // row a
                const int sca = (tca >= 4 || q >= 4) ? 0 : (tca == q ? D.sc_mch : D.sc_mis);
                int h = diag + sca;
                h = h > upE ? h : upE; h = h > fa ? h : fa;
                int hh = h - gapoe; hh = hh > 0 ? hh : 0;
                int ea = upE - gape; ea = ea > 0 ? ea : 0; ea = ea > hh ? ea : hh;
                fa -= gape; fa = fa > 0 ? fa : 0; fa = fa > hh ? fa : hh;
                if (h > rmax_a) { rmax_a = h; rarg_a = j; }
                const int ha = h;
                pubH = ha; pubE = ea;
                if (ok_b) {          // row b: the row above is row a at this column (ha, ea), its diagonal row a's cell of the step before
                    const int scb = (tcb >= 4 || q >= 4) ? 0 : (tcb == q ? D.sc_mch : D.sc_mis);
                    int g = (j >= 1 ? ha_prev : 0) + scb;
                    g = g > ea ? g : ea; g = g > fb ? g : fb;
                    int gg = g - gapoe; gg = gg > 0 ? gg : 0;
                    int eb = ea - gape; eb = eb > 0 ? eb : 0; eb = eb > gg ? eb : gg;
                    fb -= gape; fb = fb > 0 ? fb : 0; fb = fb > gg ? fb : gg;
                    if (g > rmax_b) { rmax_b = g; rarg_b = j; }
                    pubH = g; pubE = eb;
                }
                ha_prev = ha;
                if (lane == 63) { Hb[j] = pubH; Eb[j] = pubE; }
            }
        }
        __syncthreads();
        // klib's rule over the tile's rows in order (a of lane 0, b of lane 0, a of lane 1, ...): a row counts when its maximum exceeds every earlier row's, the first such
        // row that reaches endsc ends the pass.  P = the running maximum after a row, by a prefix maximum over the lanes
        {
            const int ra = ok_a ? rmax_a : INT32_MIN, rb = ok_b ? rmax_b : INT32_MIN;
            int Q = ra > rb ? ra : rb;
            for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(Q, o); if (lane >= o) Q = Q > x ? Q : x; }
            int base = __shfl_up(Q, 1);
            if (lane == 0) base = INT32_MIN;
            base = base > gmax ? base : gmax;                       // the running maximum before this lane's rows
            const int Pa = base > ra ? base : ra, Pb = Pa > rb ? Pa : rb;
            const bool hit_a = Pa > base && Pa >= endsc, hit_b = Pb > Pa && Pb >= endsc;
            const unsigned long long hits = __ballot(hit_a || hit_b);
            int M;
            if (hits) { const int Lh = __ffsll((long long)hits) - 1; const int ha_l = __shfl((int)hit_a, Lh); const int pa_l = __shfl(Pa, Lh), pb_l = __shfl(Pb, Lh); M = ha_l ? pa_l : pb_l; }
            else M = __shfl(Pb, 63);
            if (M > gmax) {
                const int F = __ffsll((long long)__ballot(Pa == M || Pb == M)) - 1;          // the first row whose running maximum is M: where it was reached
                const int is_a = __shfl((int)(Pa == M), F);
                gmax = M; te = i0 + 2 * F + (is_a ? 0 : 1);
                const int qa = __shfl(rarg_a, F), qb = __shfl(rarg_b, F);
                qe = is_a ? qa : qb;
            }
            if (hits) stop = true;
        }
    }
    if (qe < 0 && qlen > 0) qe = 0;
    score = gmax; te_out = te; qe_out = qe;
}
__device__ __attribute__((noinline)) void pe_sw_local_wave(const dp_launch_t& D, dp_lds_t& L, const moni_dp_task_t& task, moni_dp_result_t* out) {
    moni_dp_result_t r;
    r.max = 0; r.max_q = r.max_t = -1; r.mqe = 0; r.mqe_t = -1; r.mte = -1; r.mte_q = -1; r.score = 0; r.reach_end = 0; r.zdropped = 0; r.n_cigar = 0; r.cigar_off = 0;
    if (task.qlen > 0 && task.tlen > 0) {
        int score, te, qe, s2, te2, qe2;
        pe_sw_pass_wave(D, L, task, task.qlen, task.tlen, false, 0, 0, 0x10000, score, te, qe);
        pe_sw_pass_wave(D, L, task, qe + 1, task.tlen, true, qe, te, score, s2, te2, qe2);
        r.score = score; r.max_t = te; r.max_q = qe;
        if (score == s2) { r.mte = te - te2; r.mte_q = qe - qe2; }
    }
    if ((threadIdx.x & 63) == 0) *out = r;
}
// the same for the host pipeline for pairs (pe_big.h): one lane per request
extern "C" __global__ void __launch_bounds__(64) pe_sw_kernel(const dp_launch_t D, const moni_dp_task_t* tasks, uint32_t n, pe_sw_ws_t* ws, moni_dp_result_t* res) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) pe_sw_local(D, ws[i], tasks[i], res + i);
}

struct pe_slot_t {
    pe_ws_t ws;
    moni_dp_result_t res[AC_MAX_TASKS];
    uint32_t cig[AK_CIG_CAP];
    pe_sw_ws_t sw;
};

struct pe_rec_t {                            // one per pair
    uint32_t status;                         // 0 not aligned, 1 aligned (finalize: the final paired_chain_score ran), 2 overflow
    uint32_t strand;
    int32_t tot, score2, score2_m[2], sub_n, pad;
    uint32_t orphan[2];                      // the mate was placed by orphan recovery
    long long dist;
    int32_t mate_score[2];
    uint32_t filled[2];
    uint64_t ref_pos[2];
    int32_t as[2];
    uint32_t n_cigar[2], n_alt[2];
    uint64_t cigar_off[2], alt_off[2];       // into the pools
};

struct pe_args_t {
    pe_params_t PP;
    dp_launch_t D;
    const moni_mem_t* mems;
    const uint64_t* occs;
    const uint64_t* read_mem_off;
    const uint32_t* aux;
    const uint64_t* offs;
    const int32_t* min_score_of_len;
    uint32_t max_len;
    uint64_t pair_lo, n_pairs;               // this launch takes pairs [pair_lo, pair_lo + n_pairs) of the resident batch; records go to recs[pair - pair_lo]
    const uint32_t* pair_list;               // or, when set, the *n_list pairs it names (the pairs the staged kernels of pe_fast.hip hand over)
    const uint32_t* n_list;
    uint32_t nl, sw_wave;                    // sw_wave: the orphan search by the whole wave (pe_sw_local_wave) instead of the pair's own lane; nl: pairs in flight per wavefront (lanes 0 .. nl-1 run state machines): the wave solves its pairs' DP problems one after
                                             // the other, so fewer pairs per wave = shorter serial DP phases, more waves
    pe_slot_t* slots;                        // gridDim.x * nl
    ak_wave_t* waves;                        // gridDim.x
    pe_rec_t* recs;
    uint32_t* cig_pool; uint64_t cig_cap;
    moni_alt_t* alt_pool; uint64_t alt_cap;
    unsigned long long* cursors;             // [0] cigar pool, [1] alt pool, [2] DP problems run, [3] their cells, [4] next pair
};

__device__ __attribute__((noinline)) void pe_write_record(const pe_args_t& A, const pe_ws_t& S, uint64_t pair) {
    pe_rec_t R;
    R.status = S.W.overflow ? 2u : (S.W.aligned ? 1u : 0u);
    R.strand = S.strand; R.tot = S.final.tot; R.score2 = S.score2; R.sub_n = S.sub_n; R.pad = 0; R.dist = S.final.dist;
    R.mate_score[0] = S.final.m1.score; R.mate_score[1] = S.final.m2.score;
    for (int k = 0; k < 2; ++k) {
        R.score2_m[k] = S.score2_m[k];
        R.orphan[k] = 0; R.filled[k] = 0; R.ref_pos[k] = 0; R.as[k] = 0; R.n_cigar[k] = 0; R.n_alt[k] = 0; R.cigar_off[k] = 0; R.alt_off[k] = 0;
    }
    if (R.status == 1 && A.PP.finalize) {
        for (int k = 0; k < 2 && R.status == 1; ++k) {
            if (!S.filled[k]) continue;
            const unsigned long long co = atomicAdd(&A.cursors[0], (unsigned long long)S.n_cigar[k]);
            const unsigned long long ao = atomicAdd(&A.cursors[1], (unsigned long long)S.n_alt[k]);
            if (co + S.n_cigar[k] > A.cig_cap || ao + S.n_alt[k] > A.alt_cap) { R.status = 2; break; }
            R.filled[k] = 1; R.orphan[k] = S.orphan[k]; R.ref_pos[k] = S.ref_pos[k]; R.as[k] = S.as[k];
            R.n_cigar[k] = S.n_cigar[k]; R.cigar_off[k] = co; R.n_alt[k] = S.n_alt[k]; R.alt_off[k] = ao;
            for (uint32_t i = 0; i < S.n_cigar[k]; ++i) A.cig_pool[co + i] = S.cigar[k][i];
            for (uint32_t i = 0; i < S.n_alt[k]; ++i) { moni_alt_t x; x.pos = S.alt_pos[k][i]; x.score = S.alt_score[k][i]; x.pad = 0; A.alt_pool[ao + i] = x; }
        }
    }
    A.recs[pair - A.pair_lo] = R;
}

extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
pe_align_kernel(const pe_args_t A) {
    __shared__ dp_lds_t L;
    __shared__ moni_dp_task_t s_tasks[AC_MAX_TASKS];
    __shared__ unsigned long long s_cnt[2];
    const int lane = threadIdx.x;
    if (lane < 2) s_cnt[lane] = 0;
    __syncthreads();
    const int NL = (int)A.nl;
    pe_slot_t* __restrict__ S = A.slots + (size_t)blockIdx.x * NL + (lane < NL ? lane : 0);
    pe_ws_t& W = S->ws;
    uint8_t* __restrict__ dirs = A.waves[blockIdx.x].dirs;
    int state = lane < NL ? 0 : 2;             // 0: wants a pair, 1: waits for DP results, 2: no more pairs, 3: finished, record not yet written
    uint64_t pair = 0;
    while (true) {
        // ---- phase 1 (lane-private): take a pair; seeds -> chains -> first DP request ----
        const bool start = __popcll(__ballot(state == 0 || state == 3)) >= (NL + 1) / 2 || __ballot(state == 1) == 0ull;
        if (state == 3 && start) { pe_write_record(A, W, pair); state = 0; }
        if (state == 0 && start) {
            const unsigned long long nxt = atomicAdd(&A.cursors[4], 1ull);
            pair = A.pair_lo + nxt;
            if (A.pair_list) { if (nxt >= (unsigned long long)*A.n_list) state = 2; else pair = A.pair_list[nxt]; }
            else if (pair >= A.pair_lo + A.n_pairs) state = 2;
            if (state != 2) {
                bool chained = false;
                for (int k = 0; k < 2; ++k) {
                    const uint64_t r = 2 * pair + k;
                    W.off[k] = A.offs[r]; W.m[k] = (uint32_t)(A.offs[r + 1] - A.offs[r]);
                    W.min_score_m[k] = A.min_score_of_len[W.m[k] <= A.max_len ? W.m[k] : A.max_len];
                }
                W.min_score = W.min_score_m[0] + W.min_score_m[1];
                if (W.m[0] >= 4096 || W.m[1] >= 4096 || W.m[0] == 0 || W.m[1] == 0) {          // beyond the kernel's DP / the 16-bit read coordinates of the chaining nodes
                    ac_reset(W.W); W.W.overflow = (W.m[0] >= 4096 || W.m[1] >= 4096) ? 1u : 0u;
                    W.final.tot = 0; W.final.dist = 0; W.final.m1.score = W.final.m2.score = 0; W.score2 = W.score2_m[0] = W.score2_m[1] = 0; W.sub_n = 0; W.strand = 0;
                    W.filled[0] = W.filled[1] = 0; W.n_alt[0] = W.n_alt[1] = 0;
                } else chained = pe_init(W, A.PP, A.mems, A.read_mem_off, A.aux, A.occs, pair);
                if (chained) pe_drive(W, A.PP, nullptr, nullptr);
                if (chained && !W.W.overflow && W.W.stage != AC_DONE) state = 1;
                else pe_write_record(A, W, pair);
            }
        }
        const unsigned long long waiting = __ballot(state == 1);
        if (waiting == 0ull) { if (__ballot(state == 0 || state == 3) == 0ull) break; continue; }
        // the local-alignment requests of orphan recovery: each waiting lane solves its own (at most one per round)
        if (!A.sw_wave && state == 1) for (uint32_t t = 0; t < W.W.n_tasks; ++t) if (W.W.tasks[t].flag & DP_EZ_LOCAL) pe_sw_local(A.D, S->sw, W.W.tasks[t], &S->res[t]);
        __threadfence();
        // ---- phase 2 (whole wave): the DP problems of every waiting pair, one pair after the other ----
        for (unsigned long long todo = waiting; todo; todo &= todo - 1) {
            const int src = __ffsll((long long)todo) - 1;
            pe_slot_t* __restrict__ Q = A.slots + (size_t)blockIdx.x * NL + src;
            const uint32_t* __restrict__ tsrc = reinterpret_cast<const uint32_t*>(Q->ws.W.tasks);
            constexpr uint32_t TW = AC_MAX_TASKS * (uint32_t)(sizeof(moni_dp_task_t) / 4);
            const uint32_t nt = Q->ws.W.n_tasks;
            __syncthreads();
            for (uint32_t x = (uint32_t)lane; x < TW; x += 64) ((uint32_t*)s_tasks)[x] = tsrc[x];
            __syncthreads();
            bool too_big = false;
            uint32_t cig_used = 0;
            for (uint32_t t = 0; t < nt; ++t) {
                const moni_dp_task_t task = s_tasks[t];
                if (task.flag & DP_EZ_LOCAL) {                      // orphan search: by the wave, or already solved by the pair's own lane above
                    if (A.sw_wave) {
                        if (task.qlen > DP_LDS_Q) { too_big = true; break; }
                        pe_sw_local_wave(A.D, L, task, &Q->res[t]);
                        __syncthreads();
                    }
                    continue;
                }
                const bool with_cigar = !(task.flag & DP_EZ_SCORE_ONLY);
                const uint32_t cig_need = with_cigar && task.qlen > 0 && task.tlen > 0 ? (uint32_t)(task.qlen + task.tlen + 2) : 0u;
                if (task.qlen > DP_LDS_Q || task.tlen > DP_LDS_T || cig_used + cig_need > AK_CIG_CAP ||
                    (with_cigar && (uint64_t)(task.qlen + task.tlen - 1) * (uint64_t)task.tlen > AK_DIRS_CAP)) { too_big = true; break; }
                const uint32_t cig_at = cig_used;
                cig_used += cig_need;
                (void)extz_wave_lds_lite(A.D, task, L, dirs, Q->cig + cig_at, &Q->res[t], cig_at);
                if (lane == 0) { s_cnt[0]++; s_cnt[1] += (unsigned long long)(task.qlen > 0 ? task.qlen : 0) * (unsigned long long)(task.tlen > 0 ? task.tlen : 0); }
                __syncthreads();
            }
            if (lane == 0 && too_big) Q->ws.W.overflow = 1;
        }
        __threadfence();
        __syncthreads();
        // ---- phase 3 (lane-private): results -> next DP request, or the finished record ----
        if (state == 1) {
            if (!W.W.overflow) pe_drive(W, A.PP, S->res, S->cig);
            if (W.W.overflow || W.W.stage == AC_DONE) state = 3;
        }
    }
    __syncthreads();
    if (lane == 0) { atomicAdd(&A.cursors[2], s_cnt[0]); atomicAdd(&A.cursors[3], s_cnt[1]); }
}

// ------------------------------------------------------------------------------------------------------------------------------
// pe_orphan_kernel: the pairs the staged kernels hand over, in three launches, so that the chains of a pair that needs orphan recovery are scored
// by several waves at once instead of one after the other on one (pe_core.h: pe_orec_t; ~30 ms for the slowest pair of a launch otherwise, which
// nothing hides at the end of a batch).  One pair per wave, lane 0 runs its state machine, the wave solves its DP requests.
//   MODE 3  every listed pair: to its end (record written), or to the point where the orphan loop would begin - its state stays in park[li]
//   MODE 1  the parked pairs x nsplit: a copy of the parked state runs the loop for block `part` of the chains and records their scores (a memo answers requests it has solved for an earlier chain)
//   MODE 2  the parked pairs: the loop replayed from the records in chain order (no DP), the final alignments, the record
// ------------------------------------------------------------------------------------------------------------------------------
struct pe_pslot_t { pe_ws_t ws; moni_dp_result_t res[AC_MAX_TASKS]; uint32_t cig[AK_CIG_CAP]; moni_dp_result_t memo[64]; };      // memo: results of the requests the scoring pass has solved for this item
struct pe_oargs_t {
    pe_pslot_t* park; uint32_t* parked; uint32_t cap;      // the first cap listed pairs may park
    pe_orec_t* orec;                                        // cap x AC_MAX_CHAINS
    uint32_t nsplit, tag;
    pe_pslot_t* k1_slots;                                   // MODE 1: one per wave of its launch (no one-lane rows)
    uint8_t* k1_dirs; uint32_t k1_dirs_cap;                 // and direction bytes for the gap fills between anchors (small problems: a larger one sends the pair to the host pipeline)
};

// A pair's chains lie on haplotypes that mostly agree around it: the requests of one chain (the anchored mate's score-only fills, the local alignment of the other
// mate in the window, the extension over the narrowed window) are the requests of the next chain with the same query segment against a target window that spells
// the same string.  Per item of the scoring pass a memo of up to 64 solved requests: lane e holds entry e's key (query segment, lengths, flags) and target offset, the results are in the wave's slot;
// a request whose key matches and whose window equals the entry's (ak_same_target: compared by the wave) takes the entry's result.
// the wave solves the DP requests lane 0's state machine has queued in ws (results to res / cig); false: one of them is beyond the kernel
__device__ __forceinline__ bool pe_wave_solve(const pe_args_t& A, dp_lds_t& L, moni_dp_task_t* s_tasks, unsigned long long* s_cnt, uint8_t* dirs, uint64_t dirs_cap, pe_ws_t& ws,
                                              moni_dp_result_t* res, uint32_t* cig, moni_dp_result_t* memo, uint64_t& mk, uint64_t& mt, uint32_t& memo_n) {
    const int lane = threadIdx.x;
    const uint32_t* __restrict__ tsrc = reinterpret_cast<const uint32_t*>(ws.W.tasks);
    constexpr uint32_t TW = AC_MAX_TASKS * (uint32_t)(sizeof(moni_dp_task_t) / 4);
    const uint32_t nt = ws.W.n_tasks;
    __syncthreads();
    for (uint32_t x = (uint32_t)lane; x < TW; x += 64) ((uint32_t*)s_tasks)[x] = tsrc[x];
    __syncthreads();
    bool too_big = false;
    uint32_t cig_used = 0;
    for (uint32_t t = 0; t < nt; ++t) {
        const moni_dp_task_t task = s_tasks[t];
        const bool memoable = memo && (task.flag & (DP_EZ_SCORE_ONLY | DP_EZ_LOCAL)) && (task.reserved & DP_T_TEXT) && (task.reserved & DP_Q_READS) && task.qlen > 0 && task.tlen > 0 && task.tlen < 65536;
        const uint64_t key = ((task.q_off - ws.off[0]) & 0x3FFFull) | ((uint64_t)(uint32_t)task.qlen & 0xFFFull) << 14 | ((uint64_t)(uint32_t)task.tlen & 0xFFFFull) << 26 |
                             ((uint64_t)((task.flag & 0x43) | ((task.flag >> 8) & 1) << 2)) << 42 | ((uint64_t)task.reserved & 0xFFull) << 49;
        int hit = -1;
        if (memoable) {
            for (unsigned long long cand = __ballot((uint32_t)lane < memo_n && mk == key); cand && hit < 0; cand &= cand - 1) {
                const int e = __ffsll((long long)cand) - 1;
                const uint64_t toff_e = ((uint64_t)(uint32_t)__shfl((int)(mt >> 32), e) << 32) | (uint64_t)(uint32_t)__shfl((int)(mt & 0xFFFFFFFFull), e);
                if (toff_e == task.t_off || ak_same_target(A.D, (int)task.reserved, toff_e, task.t_off, task.tlen)) hit = e;
            }
        }
        if (hit >= 0) {
            if (lane == 0) res[t] = memo[hit];
            __syncthreads();
            continue;
        }
        if (task.flag & DP_EZ_LOCAL) {
            if (task.qlen > DP_LDS_Q) { too_big = true; break; }
            pe_sw_local_wave(A.D, L, task, &res[t]);
            __syncthreads();
            if (memoable && memo_n < 64) { if (lane == 0) { memo[memo_n] = res[t]; __threadfence(); } if ((uint32_t)lane == memo_n) { mk = key; mt = task.t_off; } ++memo_n; __syncthreads(); }
            continue;
        }
        const bool with_cigar = !(task.flag & DP_EZ_SCORE_ONLY);
        const uint32_t cig_need = with_cigar && task.qlen > 0 && task.tlen > 0 ? (uint32_t)(task.qlen + task.tlen + 2) : 0u;
        if (task.qlen > DP_LDS_Q || task.tlen > DP_LDS_T || cig_used + cig_need > AK_CIG_CAP ||
            (with_cigar && (uint64_t)(task.qlen + task.tlen - 1) * (uint64_t)task.tlen > dirs_cap)) { too_big = true; break; }
        const uint32_t cig_at = cig_used;
        cig_used += cig_need;
        (void)extz_wave_lds_lite(A.D, task, L, dirs, cig + cig_at, &res[t], cig_at);
        if (lane == 0) { s_cnt[0]++; s_cnt[1] += (unsigned long long)(task.qlen > 0 ? task.qlen : 0) * (unsigned long long)(task.tlen > 0 ? task.tlen : 0); }
        __syncthreads();
        if (memoable && memo_n < 64) { if (lane == 0) { memo[memo_n] = res[t]; __threadfence(); } if ((uint32_t)lane == memo_n) { mk = key; mt = task.t_off; } ++memo_n; __syncthreads(); }
    }
    return !too_big;
}

template <int MODE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) pe_orphan_kernel(const pe_args_t A, const pe_oargs_t O) {
    __shared__ dp_lds_t L;
    __shared__ moni_dp_task_t s_tasks[AC_MAX_TASKS];
    __shared__ unsigned long long s_cnt[2];
    __shared__ uint32_t s_go;
    const int lane = threadIdx.x;
    uint64_t mk = ~0ull, mt = 0; uint32_t memo_n = 0;
    if (lane < 2) s_cnt[lane] = 0;
    __syncthreads();
    pe_slot_t* const own = MODE == 1 ? nullptr : A.slots + (size_t)blockIdx.x;
    pe_pslot_t* const own1 = MODE == 1 ? O.k1_slots + (size_t)blockIdx.x : nullptr;
    uint8_t* __restrict__ dirs = MODE == 1 ? O.k1_dirs + (size_t)blockIdx.x * O.k1_dirs_cap : A.waves[blockIdx.x].dirs;
    const uint64_t dirs_cap = MODE == 1 ? O.k1_dirs_cap : AK_DIRS_CAP;
    const uint32_t n_list = *A.n_list;
    const uint32_t n_park = n_list < O.cap ? n_list : O.cap;
    const unsigned long long n_items = MODE == 3 ? (unsigned long long)n_list : MODE == 1 ? (unsigned long long)n_park * O.nsplit : (unsigned long long)n_park;
    while (true) {
        unsigned long long nxt = 0;
        if (lane == 0) nxt = atomicAdd(&A.cursors[MODE == 3 ? 4 : MODE == 1 ? 5 : 6], 1ull);
        nxt = ((unsigned long long)(uint32_t)__shfl((int)(nxt >> 32), 0) << 32) | (uint32_t)__shfl((int)(nxt & 0xFFFFFFFFull), 0);
        if (nxt >= n_items) break;
        const uint32_t li = MODE == 1 ? (uint32_t)(nxt / O.nsplit) : (uint32_t)nxt, part = MODE == 1 ? (uint32_t)(nxt % O.nsplit) : 0u;
        const uint64_t pair = A.pair_list[li];
        if (MODE != 3 && !O.parked[li]) continue;
        memo_n = 0;
        const bool in_park = MODE == 2 || (MODE == 3 && li < O.cap);
        pe_ws_t* const W = in_park ? &O.park[li].ws : MODE == 1 ? &own1->ws : &own->ws;
        moni_dp_result_t* const res = in_park ? O.park[li].res : MODE == 1 ? own1->res : own->res;
        uint32_t* const cig = in_park ? O.park[li].cig : MODE == 1 ? own1->cig : own->cig;
        if (MODE == 1) {          // the parked state, copied: this wave's own loop
            const uint4* __restrict__ src = reinterpret_cast<const uint4*>(&O.park[li].ws);
            uint4* __restrict__ dst = reinterpret_cast<uint4*>(W);
            for (uint32_t x = (uint32_t)lane; x < (uint32_t)(sizeof(pe_ws_t) / sizeof(uint4)); x += 64) dst[x] = src[x];
            __threadfence();
        }
        __syncthreads();
        if (lane == 0) {
            if (MODE == 3) {
                bool chained = false;
                for (int k = 0; k < 2; ++k) {
                    const uint64_t r = 2 * pair + k;
                    W->off[k] = A.offs[r]; W->m[k] = (uint32_t)(A.offs[r + 1] - A.offs[r]);
                    W->min_score_m[k] = A.min_score_of_len[W->m[k] <= A.max_len ? W->m[k] : A.max_len];
                }
                W->min_score = W->min_score_m[0] + W->min_score_m[1];
                W->o_mode = 0; W->o_parked = 0;
                if (W->m[0] >= 4096 || W->m[1] >= 4096 || W->m[0] == 0 || W->m[1] == 0) {
                    ac_reset(W->W); W->W.overflow = (W->m[0] >= 4096 || W->m[1] >= 4096) ? 1u : 0u;
                    W->final.tot = 0; W->final.dist = 0; W->final.m1.score = W->final.m2.score = 0; W->score2 = W->score2_m[0] = W->score2_m[1] = 0; W->sub_n = 0; W->strand = 0;
                    W->filled[0] = W->filled[1] = 0; W->n_alt[0] = W->n_alt[1] = 0;
                } else chained = pe_init(*W, A.PP, A.mems, A.read_mem_off, A.aux, A.occs, pair);
                if (chained) { W->o_mode = in_park ? 3u : 0u; pe_drive(*W, A.PP, nullptr, nullptr); }
                else W->W.stage = AC_DONE;
            } else {
                W->o_mode = (uint32_t)MODE; W->o_part = part; W->o_nsplit = O.nsplit; W->o_tag = O.tag; W->o_parked = 0;
                W->orec = O.orec + (size_t)li * AC_MAX_CHAINS;
                pe_drive(*W, A.PP, nullptr, nullptr);
            }
            s_go = (!W->W.overflow && W->W.stage != AC_DONE && !W->o_parked) ? 1u : 0u;
        }
        __syncthreads();
        while (s_go) {
            __threadfence();
            const bool ok = pe_wave_solve(A, L, s_tasks, s_cnt, dirs, dirs_cap, *W, res, cig, MODE == 1 ? own1->memo : nullptr, mk, mt, memo_n);
            __threadfence();
            __syncthreads();
            if (lane == 0) {
                if (!ok) W->W.overflow = 1;
                else pe_drive(*W, A.PP, res, cig);
                s_go = (!W->W.overflow && W->W.stage != AC_DONE && !W->o_parked) ? 1u : 0u;
            }
            __syncthreads();
        }
        if (lane == 0) {
            if (MODE == 3) {
                const bool parked = in_park && W->o_parked && !W->W.overflow;
                if (li < O.cap) O.parked[li] = parked ? 1u : 0u;
                if (!parked) pe_write_record(A, *W, pair);
            } else if (MODE == 1) {
                if (W->W.overflow && W->W.stage >= PE_O_LOOP && W->W.i < AC_MAX_CHAINS) { pe_orec_t& R = W->orec[W->W.i]; R.kind = 2; R.tag = O.tag; }      // the replay stops here: status 2
            } else pe_write_record(A, *W, pair);
        }
        __syncthreads();
    }
    __syncthreads();
    if (lane == 0) { atomicAdd(&A.cursors[2], s_cnt[0]); atomicAdd(&A.cursors[3], s_cnt[1]); }
}
