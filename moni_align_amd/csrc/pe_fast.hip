// The paired-end path as staged kernels (the common case of aligner::align(paired_alignment_t&), include/aligner/aligner_ksw2.hpp:1000-1326),
// built on the single-end stages of align_fast.hip.  pe_align_kernel (pe_kernel.hip) runs every pair as a lane-private state machine out of a
// 159 KB slot in HBM with the wave solving its pairs' DP problems one after the other; here each phase has the mapping that suits it:
//
//   pe_plan_kernel     one wavefront per PAIR, state in LDS: the pair's four (mate, strand) seed lists in the reference's order with r_offset
//                      (aligner_ksw2.hpp:1012-1040), direction filter, frequency filter, anchors, the mate-aware chaining of af_chain (chain.hpp:221-438),
//                      get_best_scores' loop over the chains (aligner_ksw2.hpp:1329-1400; which chains it scores does not depend on any score) with
//                      check_paired_left_MEM (:1471-1534), and for every paired chain it scores the DP problems of BOTH mates' fill_chain
//                      (paired_chain_score, :2115-2170): one af_cand_t per mate, its share of the chain's anchors, mate in bit 0 of `pad`
//   bin_tasks / af_chunk / dp_lane / global_task kernels: as for single reads (a plan belongs to a pair: af_args_t::pe)
//   pe_select_kernel   one lane per pair: both mates' chain_score arithmetic, the pairing term, check_max_score per mate, best_scores, sub_n, the
//                      final chain and which mates get a CIGAR; queues the tracebacks
//   traceback_kernel   as for single reads
//   pe_finish_kernel   one lane per pair: both stitched CIGARs and the pe_rec_t record the host finishing (pe_host.hpp) turns into two SAM lines
//
// What is outside the common case goes to pe_align_kernel over a list, unchanged: pairs that fail jointly and need orphan recovery
// (aligner_ksw2.hpp:1536-1640), more seeds / anchors / chains than the LDS arrays hold, more than 8 paired chains to score, overlapping anchors whose
// window is beyond the global tile, wildcard bases in a DP operand, an extension short of the query end.  Results are identical either way.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// (included by moni_hip.hip after align_fast.hip and pe_kernel.hip)

#define PEF_MAX_Q 8                   // paired chains scored per pair (two af_cand_t each)
#define PEF_RAW 160                   // seeds of both mates staged per pair

enum { PEF_ST_UNALIGNED = 0, PEF_ST_ALIGNED = 1, PEF_ST_FALLBACK = 2 };

struct pe_sel_t {                     // what pe_select_kernel leaves for pe_finish_kernel
    uint32_t status, strand;
    int32_t tot, score2, score2_m[2], sub_n;
    long long dist;
    int32_t mate_score[2];
    uint32_t fill_on[2], final_q;
    uint32_t tb0[2], n_tb[2];
    uint64_t ref_pos[2]; int32_t as[2];
    uint32_t n_alt[2];
    uint64_t alt_pos[2][PEF_MAX_Q]; int32_t alt_score[2][PEF_MAX_Q];
};

struct pef_args_t {
    af_args_t G;                      // G.pe = 1; G.A.read_lo = first pair of the launch, G.A.n_reads = its pairs; G.A.offs / mems / occs / read_mem_off: of the 2 N reads
    pe_params_t PP;
    const uint32_t* aux;              // the seeding kernels' marks of MEM halves
    pe_rec_t* recs;                   // one per pair of the launch
    uint32_t* cig_pool; uint64_t cig_cap;
    moni_alt_t* alt_pool; uint64_t alt_cap;
    unsigned long long* cursors;      // [0] cigar pool, [1] alt pool
};

typedef af_wave_tt<AF_MAX_ANCH, AF_MAX_CHAINS, AF_MAX_MEMS, 2 * PEF_MAX_Q, AF_PLAN_AN, AF_MAX_TASKS_READ, 1> pef_wave_t;
// most pairs: 2 x 150 bp on a 13-sequence index seed ~8 MEMs and ~90 anchors per pair (p99.9: 13 and 157; profiles/r03l/pe_seed_hist.txt): 9 KB of LDS, 4 waves per SIMD.
// A pair that overflows this instance is put on a list for the large one (LEVEL 1).
typedef af_wave_tt<160, 80, 24, 2 * PEF_MAX_Q, 64, 48, 1> pef_wave_small_t;
// -Z (find_chains_secondary): the same two instances with the second track's arrays
typedef af_wave_tt<160, 160, 24, 2 * PEF_MAX_Q, 64, 48, 1, 1> pef_wave_small_z_t;          // (as many chains as anchors: the second track starts one at about every anchor, and with 80 nearly half of the benchmark's pairs ran here and again on the large instance - profiles/r05e)
typedef af_wave_tt<AF_MAX_ANCH, AF_MAX_CHAINS, AF_MAX_MEMS, 2 * PEF_MAX_Q, AF_PLAN_AN, AF_MAX_TASKS_READ, 1, 1> pef_wave_z_t;
// -Z: the second track starts a chain at about every anchor of a repeat-rich pair - more chains than AF_MAX_CHAINS for one pair in eighty of the benchmark
// (profiles/r04p: 12 731 of 1 M pairs went to pe_align_kernel for it, two thirds of the -Z run's time).  LEVEL 2: four times the chains (and twice the anchors).
typedef af_wave_tt<2 * AF_MAX_ANCH, 4 * AF_MAX_CHAINS, AF_MAX_MEMS, 2 * PEF_MAX_Q, AF_PLAN_AN, AF_MAX_TASKS_READ, 1, 1> pef_wave_zz_t;          // (the sorts' scratch arrays are per anchor: chains <= anchors)
#define PEF_RAW_SMALL 32

// ------------------------------------------------------------------------------------------------------------------------------
// pe_plan_kernel
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void iota_kernel(uint32_t* __restrict__ out, uint32_t n) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = i; }

// LEVEL 0: every pair of the launch, the small instance; a pair beyond its capacities goes to G.big_list.  LEVEL 1: that list, the large instance;
// a pair beyond ITS capacities goes to pe_align_kernel - or (LAST = 0: -Z) one with more chains than it holds to G.huge_list for LEVEL 2.
template <class WT, int OCC, int RAW, int LEVEL, int LAST = 1>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) pe_plan_kernel(const pef_args_t X) {
    __shared__ WT L;
    __shared__ af_mem_t raw[RAW];               // the seeds of both mates as the seeding kernels left them (mate field: their strand bit)
    __shared__ uint32_t raw_aux[RAW];
    __shared__ uint16_t ord[RAW];               // the pair's seed list in the reference's order: raw index | call << 12
    __shared__ uint32_t sh[8];
    const af_args_t& G = X.G;
    const ak_args_t& A = G.A;
    const ac_params_t& P = A.P;
    const int lane = threadIdx.x;
    const uint32_t n_work = LEVEL == 0 ? (uint32_t)A.n_reads : G.ctr[LEVEL == 1 ? AFC_BIG : AFC_HUGE];
    while (true) {
        uint32_t w_in = 0;
        if (lane == 0) w_in = atomicAdd(&G.ctr[LEVEL == 0 ? AFC_READ_CUR : LEVEL == 1 ? AFC_BIG_CUR : AFC_HUGE_CUR], 1u);
        w_in = (uint32_t)__shfl((int)w_in, 0);
        if (w_in >= n_work) break;
        const uint32_t r_in = LEVEL == 0 ? w_in : LEVEL == 1 ? G.big_list[w_in] : G.huge_list[w_in];
        const uint64_t pair = A.read_lo + r_in, r1 = 2 * pair, r2 = r1 + 1;
        const uint64_t off1 = A.offs[r1], off2 = A.offs[r2];
        const uint32_t m1 = (uint32_t)(off2 - off1), m2 = (uint32_t)(A.offs[r2 + 1] - off2);
        const uint64_t a1 = A.read_mem_off[r1], b1 = A.read_mem_off[r1 + 1], b2 = A.read_mem_off[r2 + 1];      // seeds of mate 1: [a1, b1), of mate 2: [b1, b2)
        const uint32_t n1 = (uint32_t)(b1 - a1), n2 = (uint32_t)(b2 - b1);
        uint32_t status = AF_ST_UNALIGNED;
        const bool too_long = m1 >= AF_MAX_READ || m2 >= AF_MAX_READ;
        bool fallback = too_long || n1 + n2 > (uint32_t)RAW;
        bool retry = false;                                       // LEVEL 0: beyond this instance, not (yet) beyond the staged kernels
        bool chains_ovf = false;                                  // more chains than the instance holds
        if (m1 == 0 || m2 == 0) fallback = false;                 // (an empty mate: the pair is not aligned, pe_align_kernel's rule)
        uint32_t n_mems = 0, na = 0;
        float avg = 0.f;
        __syncthreads();
        if (!fallback && m1 > 0 && m2 > 0 && n1 + n2 > 0) {
            for (uint32_t k = lane; k < n1 + n2; k += 64) {
                const moni_mem_t g = A.mems[a1 + k];
                af_mem_t x; x.occ_off = g.occ_off; x.nocc = g.occ_cnt; x.len = (uint16_t)g.len; x.idx = (uint16_t)g.idx; x.rpos = (uint16_t)g.rpos; x.mate = (uint8_t)g.mate; x.pad = 0;
                raw[k] = x; raw_aux[k] = X.aux[a1 + k];
            }
            __syncthreads();
            if (lane == 0) {
                // the four find_mems calls (aligner_ksw2.hpp:1012-1040; pe_core.h pe_init): (mate, strand bit of the seeds, mate flags, r_offset)
                const uint32_t c_read[4] = {0u, X.PP.filter_dir ? 1u : 0u, 1u, X.PP.filter_dir ? 0u : 1u};
                const uint32_t c_rc[4] = {0u, 2u, 0u, 2u};
                uint32_t no = 0, n_dir1 = 0, n_dir2 = 0;
                bool ovf = false;
                for (int k = 0; k < 4; ++k) {                            // the MEMs themselves
                    const uint32_t base = c_read[k] ? n1 : 0u, cnt = c_read[k] ? n2 : n1;
                    for (uint32_t g = 0; g < cnt; ++g) {
                        const uint32_t ax = raw_aux[base + g];
                        if (ax >= 0xFFFFFFFDu && ax != 0xFFFFFFFFu) break;            // the halves follow all MEMs of the read
                        if (raw[base + g].mate != c_rc[k]) continue;
                        if (no >= (uint32_t)RAW) { ovf = true; break; }
                        ord[no++] = (uint16_t)((base + g) | ((uint32_t)k << 12));
                        if (k < 2) ++n_dir1; else ++n_dir2;
                    }
                }
                for (int k = 0; k < 4 && !ovf; ++k) {                    // populate_seeds: the halves of every long MEM, in the order of the MEMs
                    const uint32_t base = c_read[k] ? n1 : 0u, cnt = c_read[k] ? n2 : n1;
                    for (uint32_t g = 0; g < cnt; ++g) {
                        const uint32_t ax = raw_aux[base + g];
                        if (ax >= 0xFFFFFFFDu && ax != 0xFFFFFFFFu) break;
                        if (raw[base + g].mate != c_rc[k] || ax == 0xFFFFFFFFu) continue;
                        if (no + 2 > (uint32_t)RAW || base + ax + 1 >= n1 + n2) { ovf = true; break; }
                        ord[no++] = (uint16_t)((base + ax) | ((uint32_t)k << 12));
                        ord[no++] = (uint16_t)((base + ax + 1) | ((uint32_t)k << 12));
                    }
                }
                uint32_t lo = 0, hi = no;
                if (X.PP.filter_dir && !ovf) {                           // aligner_ksw2.hpp:1042-1100: only the two plain averages decide
                    double s1 = 0.0, s2 = 0.0;
                    for (uint32_t i = 0; i < n_dir1; ++i) s1 += (double)raw[ord[i] & 0xFFFu].len;
                    for (uint32_t i = n_dir1; i < no; ++i) s2 += (double)raw[ord[i] & 0xFFFu].len;
                    if (n_dir1 > 0) s1 = s1 / (double)n_dir1;
                    if (n_dir2 > 0) s2 = s2 / (double)n_dir2;
                    if (s1 > s2 && (s1 - s2) > X.PP.dir_thr) hi = n_dir1;
                    if (s2 > s1 && (s2 - s1) > X.PP.dir_thr) lo = n_dir1;
                }
                // seed_freq_filter over what is left, then the wave's seed list
                size_t total = 0;
                for (uint32_t i = lo; i < hi; ++i) total += raw[ord[i] & 0xFFFu].nocc;
                uint32_t k = 0; unsigned long long tot_len = 0, n_anch = 0;
                for (uint32_t i = lo; i < hi && !ovf; ++i) {
                    const af_mem_t g = raw[ord[i] & 0xFFFu];
                    if (P.filter_freq) { const double fr = static_cast<double>(g.nocc) / total; if (fr > P.freq_thr) continue; }
                    if (k >= (uint32_t)WT::MM) { ovf = true; break; }
                    const uint32_t call = ord[i] >> 12;
                    // mate flags and r_offset of the call (filter_dir: m1 F, m2 RC + |m1|, m2 F, m1 RC + |m2|; else m1 F, m1 RC + |m2|, m2 F, m2 RC + |m1|)
                    const uint32_t is_m2 = c_read[call], rc = c_rc[call];
                    af_mem_t x = g;
                    x.mate = (uint8_t)(is_m2 | rc);
                    x.rpos = (uint16_t)(g.rpos + (rc ? (is_m2 ? m1 : m2) : 0u));
                    L.mem[k++] = x;
                    tot_len += (unsigned long long)g.len * g.nocc; n_anch += g.nocc;
                }
                if (n_anch > (unsigned long long)WT::MA) ovf = true;
                sh[0] = ovf ? 0xFFFFFFFFu : k; sh[1] = (uint32_t)n_anch;
                sh[2] = __float_as_uint(n_anch ? (float)(size_t)tot_len / (size_t)n_anch : 0.f);
            }
            __syncthreads();
            if (sh[0] == 0xFFFFFFFFu) fallback = true;
            else { n_mems = sh[0]; na = sh[1]; avg = __uint_as_float(sh[2]); }
            if (!fallback && na > 0) {
                uint32_t base = 0;
                for (uint32_t i = 0; i < n_mems; ++i) {              // populate_anchors (chain.hpp:83-95): mem by mem, occurrence by occurrence
                    const af_mem_t mi = L.mem[i];
                    for (uint32_t j = lane; j < mi.nocc; j += 64) L.anch[base + j] = (A.occs[mi.occ_off + j] + mi.len - 1) | ((uint64_t)i << 40);
                    base += mi.nocc;
                }
                for (uint32_t i = lane; i < na; i += 64) { L.f[i] = 0; L.msc[i] = 0; L.p[i] = 0; L.t[i] = 0; }
            }
        }
        __syncthreads();
        if (!fallback && na > 0) {
            status = af_chain(G, L, na, avg, af_grp_t<64>(), WT::SEC && X.PP.secondary_chains != 0);
            status = (uint32_t)__shfl((int)status, 0);
            __syncthreads();
            if (status == 0xFFu) { fallback = true; chains_ovf = true; status = AF_ST_UNALIGNED; }
            else if (status == AF_ST_CAND) {
                // check_paired_left_MEM's coordinates of every chain: the leftmost anchor of each mate, lifted (aligner_ksw2.hpp:1471-1500); 0: none
                for (uint32_t ci = lane; ci < L.n_chains_sh; ci += 64) {
                    const af_chain_t ch = L.chains[ci];
                    uint64_t ref[2] = {0, 0}; bool have[2] = {false, false};
                    for (uint32_t k = 0; k < ch.cnt && !(have[0] && have[1]); ++k) {
                        const uint64_t aw = L.anch[L.pool[ch.off + ch.cnt - 1 - k]];
                        const af_mem_t ml = L.mem[aw >> 40];
                        const uint32_t mt = ml.mate & 1u;
                        if (!have[mt]) { have[mt] = true; ref[mt] = ac_seq_off(P, ac_lift(P, AF_X(aw) - ml.len + 1)) + 1; }
                    }
                    L.left_ref[ci] = ref[0]; L.left_ref2[ci] = ref[1];
                }
                __syncthreads();
                auto& PL = L.plan;
                if (lane == 0) {                                     // get_best_scores' loop ahead of its scores (aligner_ksw2.hpp:1329-1368)
                    const uint32_t n_chains = L.n_chains_sh;
                    PL.n_chains = (uint16_t)n_chains; PL.n_cand = 0; PL.n_an = 0;
                    L.n_tasks = 0;
                    uint32_t st = AF_ST_CAND;
                    int64_t* diff = L.diff; uint32_t n_diff = 0, n_left = 0;
                    for (uint32_t ci = 0; ci < n_chains && n_diff < P.check_k; ++ci) {
                        const af_chain_t ch = L.chains[ci];
                        { bool f = false; for (uint32_t q = 0; q < n_diff; ++q) f = f || diff[q] == (int64_t)ch.score; if (!f) diff[n_diff++] = ch.score; }
                        if (P.left_mem_check) {
                            bool seen = false;
                            for (uint32_t k = 0; k < n_left; ++k) {
                                const uint32_t o = L.left_idx[k];
                                const uint64_t d1 = L.left_ref[o] > L.left_ref[ci] ? L.left_ref[o] - L.left_ref[ci] : L.left_ref[ci] - L.left_ref[o];
                                const uint64_t d2 = L.left_ref2[o] > L.left_ref2[ci] ? L.left_ref2[o] - L.left_ref2[ci] : L.left_ref2[ci] - L.left_ref2[o];
                                if (d1 < P.region_dist && d2 < P.region_dist && L.chains[o].score == ch.score) seen = true;
                            }
                            if (seen) continue;
                            L.left_idx[n_left++] = (uint16_t)ci;
                        }
                        if (n_diff >= P.check_k) continue;
                        if (!((ch.mate >> 8) & 1u)) continue;            // not paired: paired_chain_score returns the empty score (aligner_ksw2.hpp:2145), nothing to compute
                        if (PL.n_cand + 2 > (uint32_t)WT::NC) { st = AF_ST_FALLBACK; atomicAdd(&G.ctr[AFC_WHY + AF_WHY_CANDS], 1u); break; }
                        for (uint32_t k = 0; k < 2; ++k) {               // one record per mate; the anchors are counted by the lanes below
                            af_cand_t& C = PL.cand[PL.n_cand++];
                            C.chain_score = ch.score; C.chain_idx = (uint16_t)ci; C.n_an = 0; C.task0 = 0; C.has_lc = C.has_rc = C.n_gap_tasks = 0; C.overlap = 0; C.score = 0; C.gtask = 0;
                            C.strand = 0; C.an0 = 0; C.pad = (uint16_t)k; C.pad2 = 0;
                        }
                    }
                    L.status_sh = st;
                }
                __syncthreads();
                status = L.status_sh;
                if (status == AF_ST_CAND) {
                    const uint32_t n_cand = PL.n_cand;
                    // every lane its mate's share of its chain: count the anchors, prefix sum, write them, then the problems (count, prefix sum, write)
                    uint32_t cnt = 0;
                    if ((uint32_t)lane < n_cand) {
                        const af_chain_t ch = L.chains[PL.cand[lane].chain_idx];
                        for (uint32_t k = 0; k < ch.cnt; ++k) cnt += (L.mem[L.anch[L.pool[ch.off + k]] >> 40].mate & 1u) == (uint32_t)(PL.cand[lane].pad & 1u) ? 1u : 0u;
                    }
                    uint32_t incl = cnt;
                    for (int o = 1; o < 2 * PEF_MAX_Q; o <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += x; }
                    const uint32_t total_an = n_cand ? (uint32_t)__shfl((int)incl, (int)n_cand - 1) : 0u;
                    const bool an_bad = __ballot((uint32_t)lane < n_cand && (cnt == 0 || cnt > 255u)) != 0ull || total_an > (uint32_t)WT::NA;
                    uint32_t nt = 0;
                    if (!an_bad && (uint32_t)lane < n_cand) {
                        af_cand_t& C = PL.cand[lane];
                        C.an0 = (uint16_t)(incl - cnt); C.n_an = (uint8_t)cnt;
                        af_cand_anchors(L, C, 1u << (C.pad & 1u), 0u);
                        const uint32_t k = C.pad & 1u;
                        nt = af_build_cand<false>(G, L, C, k ? off2 : off1, k ? m2 : m1, 0u);
                    }
                    const bool bad = an_bad || __ballot(nt == 0xFFFFFFFFu) != 0ull;
                    uint32_t ti = nt;
                    for (int o = 1; o < 2 * PEF_MAX_Q; o <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)ti, o); if (lane >= o) ti += x; }
                    const uint32_t total_t = n_cand ? (uint32_t)__shfl((int)ti, (int)n_cand - 1) : 0u;
                    if (LEVEL == 0 && (bad || total_t > (uint32_t)WT::NT)) { retry = true; status = AF_ST_UNALIGNED; }
                    else if (bad || total_t > (uint32_t)WT::NT) {
                        status = AF_ST_FALLBACK;
                        if (lane == 0) atomicAdd(&G.ctr[AFC_WHY + (an_bad ? AF_WHY_CHAIN_LEN : bad ? AF_WHY_TASK_SIZE : AF_WHY_CAPACITY)], 1u);
                    } else {
                        if ((uint32_t)lane < n_cand) {
                            af_cand_t& C = PL.cand[lane];
                            const uint32_t k = C.pad & 1u;
                            C.task0 = ti - nt;
                            af_build_cand<true>(G, L, C, k ? off2 : off1, k ? m2 : m1, ti - nt);
                        }
                        if (lane == 0) { L.n_tasks = total_t; PL.n_an = total_an; }
                    }
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        if (LEVEL == 0 && fallback && !too_long) retry = true;
        if (LEVEL == 1 && !LAST && chains_ovf) retry = true;
        if (retry) {                                              // the next larger instance takes the pair
            if (lane == 0) { if (LEVEL == 0) G.big_list[atomicAdd(&G.ctr[AFC_BIG], 1u)] = r_in; else G.huge_list[atomicAdd(&G.ctr[AFC_HUGE], 1u)] = r_in; }
            __syncthreads();
            continue;
        }
        if (fallback) { status = AF_ST_FALLBACK; if (lane == 0) atomicAdd(&G.ctr[AFC_WHY + (too_long ? AF_WHY_LONG : chains_ovf ? AF_WHY_CHAINS : AF_WHY_ANCHORS)], 1u); }
        // ---- the pair's tasks go to its own slots; the plan goes to HBM ----
        auto& PL = L.plan;
        const uint32_t nt = status == AF_ST_CAND ? L.n_tasks : 0u;
        const uint32_t t0 = r_in * AF_MAX_TASKS_READ;
        if (status == AF_ST_CAND) {
            if ((uint32_t)lane < nt) G.tasks[t0 + lane] = PL.tasks[lane];
            if ((uint32_t)lane < PL.n_cand) PL.cand[lane].task0 += t0;
        }
        if (lane == 0) {
            G.ntasks[r_in] = (uint8_t)nt;
            if (status != AF_ST_CAND) { PL.n_cand = 0; PL.n_chains = status == AF_ST_UNALIGNED ? 0 : PL.n_chains; }
            PL.status = (uint8_t)status; PL.final_cand = 0; PL.n_alt = 0; PL.pad = 0; PL.score2 = 0; PL.ref_pos = PL.ref_len = 0; PL.tb0 = 0; PL.pad2 = 0;
            PL.min_score = 0;
            if (status == AF_ST_FALLBACK) G.fb_list[atomicAdd(G.fb_n, 1u)] = (uint32_t)pair;
        }
        __syncthreads();
        {
            const uint32_t words = (uint32_t)((offsetof(af_plan_t, cand) + (status == AF_ST_CAND ? PL.n_cand : 0u) * sizeof(af_cand_t)) / 4);
            const uint32_t* src = reinterpret_cast<const uint32_t*>(&PL);
            uint32_t* dst = reinterpret_cast<uint32_t*>(G.plans + r_in);
            for (uint32_t w = lane; w < words; w += 64) dst[w] = src[w];
            const uint32_t awords = status == AF_ST_CAND ? PL.n_an * (uint32_t)(sizeof(af_anchor_t) / 4) : 0u;
            const uint32_t* asrc = reinterpret_cast<const uint32_t*>(PL.an);
            uint32_t* adst = reinterpret_cast<uint32_t*>(G.plans[r_in].an);
            for (uint32_t w = lane; w < awords; w += 64) adst[w] = asrc[w];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// pe_select_kernel: get_best_scores with the DP results in hand (aligner_ksw2.hpp:1368-1431), the decision of align(paired_alignment_t&) (:1244-1326)
// ------------------------------------------------------------------------------------------------------------------------------
struct pef_best_t { int32_t tot; int32_t s[2]; uint64_t pos[2], lft[2]; long long dist; uint32_t q; };

__global__ void __launch_bounds__(256) pe_select_kernel(const pef_args_t X) {
    const af_args_t& G = X.G;
    const ak_args_t& A = G.A;
    const ac_params_t& P = A.P;
    const uint64_t r_in = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r_in >= A.n_reads) return;
    af_plan_t& PL = G.plans[r_in];
    pe_sel_t& S = G.pe_sel[r_in];
    const uint64_t pair = A.read_lo + r_in;
    const uint32_t m[2] = {(uint32_t)(A.offs[2 * pair + 1] - A.offs[2 * pair]), (uint32_t)(A.offs[2 * pair + 2] - A.offs[2 * pair + 1])};
    int32_t min_m[2];
    for (int k = 0; k < 2; ++k) min_m[k] = A.min_score_of_len[m[k] <= A.max_len ? m[k] : A.max_len];
    const int32_t min_score = min_m[0] + min_m[1];
    S.status = PEF_ST_UNALIGNED; S.strand = 0; S.tot = 0; S.score2 = 0; S.score2_m[0] = S.score2_m[1] = 0; S.sub_n = 0; S.dist = 0; S.mate_score[0] = S.mate_score[1] = 0;
    S.fill_on[0] = S.fill_on[1] = 0; S.final_q = 0; S.tb0[0] = S.tb0[1] = 0; S.n_tb[0] = S.n_tb[1] = 0; S.ref_pos[0] = S.ref_pos[1] = 0; S.as[0] = S.as[1] = 0; S.n_alt[0] = S.n_alt[1] = 0;
    if (PL.status == AF_ST_FALLBACK) { S.status = PEF_ST_FALLBACK; return; }
    if (PL.status != AF_ST_CAND) return;
    const uint32_t nq = PL.n_cand / 2;
    pef_best_t best[PEF_MAX_Q + 2];
    uint32_t n_best = 0;
    int32_t max_m[2] = {0, 0};
    bool fallback = false;
    uint32_t why = AF_WHY_WILDCARD;
    int32_t cs[PEF_MAX_Q][2]; uint64_t cpos[PEF_MAX_Q][2], clen[PEF_MAX_Q][2];
    for (uint32_t q = 0; q < nq && !fallback; ++q) {
        pef_best_t sc; sc.q = q;
        for (uint32_t k = 0; k < 2; ++k) {                           // chain_score of mate k's share (select_kernel's arithmetic)
            af_cand_t& C = PL.cand[2 * q + k];
            const af_anchor_t* const AN = PL.an + C.an0;
            uint32_t t = C.task0;
            int score_lc = 0, score_rc = 0, lc_t = -1, rc_t = -1;
            if (C.has_lc) { const af_res_t R = G.res[t++]; fallback |= R.flags != 0; score_lc = R.mqe; lc_t = R.mqe_t; }
            if (C.has_rc) { const af_res_t R = G.res[t++]; fallback |= R.flags != 0; score_rc = R.mqe; rc_t = R.mqe_t; }
            uint64_t ref_pos, ref_len;
            af_window(C, AN, m[k], lc_t, rc_t, ref_pos, ref_len);
            int32_t score;
            if (C.overlap) {
                score = INT32_MIN;
                if (C.gtask != ~0u) { const af_res_t R = G.res[C.gtask]; fallback |= R.flags != 0; score = R.score; }
            } else {
                uint32_t s2 = (uint32_t)score_lc + (uint32_t)score_rc;
                for (uint32_t a = 1; a < C.n_an; ++a) {
                    const af_anchor_t ap = AN[a - 1];
                    int32_t gs = ap.gap_val;
                    if (ap.gap_kind == AF_GAP_INS) gs = af_ins_score(P, (uint64_t)(uint16_t)ap.gap_val);
                    else if (ap.gap_kind == AF_GAP_TASK) { const af_res_t R = G.res[t++]; fallback |= R.flags != 0; gs = R.score; }
                    s2 += (uint32_t)((uint64_t)ap.len * (uint64_t)(int64_t)P.smatch + (uint64_t)(int64_t)gs);
                }
                s2 += (uint32_t)((uint64_t)AN[C.n_an - 1].len * (uint64_t)(int64_t)P.smatch);
                score = (int32_t)s2;
            }
            if (!ac_valid(P, ref_pos, ref_len)) score = INT32_MIN;
            C.score = score;
            cs[q][k] = score; cpos[q][k] = ref_pos; clen[q][k] = ref_len;
            sc.s[k] = score; sc.pos[k] = ref_pos; sc.lft[k] = ac_lift(P, ref_pos);
        }
        sc.dist = (long long)pe_dist(sc.pos[1], sc.pos[0] + (uint64_t)m[0]);
        sc.tot = pe_pair_total(X.PP, sc.s[0], sc.s[1], sc.dist);
        for (int k = 0; k < 2; ++k) {                                // check_max_score per mate (aligner_ksw2.hpp:528-548)
            if (sc.s[k] > max_m[k]) { max_m[k] = sc.s[k]; S.n_alt[k] = 0; }
            else if (sc.s[k] == max_m[k]) { S.alt_pos[k][S.n_alt[k]] = sc.pos[k]; S.alt_score[k][S.n_alt[k]] = sc.s[k]; ++S.n_alt[k]; }      // at most nq - 1 entries
        }
        if (sc.tot >= min_score) {                                   // the best_scores update (aligner_ksw2.hpp:1376-1400)
            bool replaced = false;
            for (uint32_t j = 0; j < n_best; ++j) {
                if (pe_dist(best[j].lft[0], sc.lft[0]) < P.region_dist && pe_dist(best[j].lft[1], sc.lft[1]) < P.region_dist) {
                    if (sc.tot > best[j].tot) {
                        if (replaced) { best[j].tot = 0; best[j].s[0] = best[j].s[1] = 0; best[j].pos[0] = best[j].pos[1] = best[j].lft[0] = best[j].lft[1] = 0; best[j].dist = 0; best[j].q = ~0u; }
                        else { best[j] = sc; replaced = true; }
                    } else { j = n_best; replaced = true; }
                }
            }
            if (!replaced) best[n_best++] = sc;
        }
    }
    if (fallback) { PL.status = AF_ST_FALLBACK; S.status = PEF_ST_FALLBACK; G.fb_list[atomicAdd(G.fb_n, 1u)] = (uint32_t)pair; atomicAdd(&G.ctr[AFC_WHY + why], 1u); return; }
    while (n_best < 2) { pef_best_t& z = best[n_best++]; z.tot = 0; z.s[0] = z.s[1] = 0; z.pos[0] = z.pos[1] = z.lft[0] = z.lft[1] = 0; z.dist = 0; z.q = ~0u; }
    // std::sort(greater) of at most PEF_MAX_Q + 2 <= 16 elements: libstdc++'s insertion sort, i.e. a stable sort (paired_score_t::operator>: tot, m1.lft, m2.lft)
    for (uint32_t i = 1; i < n_best; ++i) {
        const pef_best_t v = best[i];
        uint32_t k = i;
        auto gt = [](const pef_best_t& x, const pef_best_t& y) { return x.tot > y.tot || (x.tot == y.tot && x.lft[0] > y.lft[0]) || (x.tot == y.tot && x.lft[0] == y.lft[0] && x.lft[1] > y.lft[1]); };
        while (k > 0 && gt(v, best[k - 1])) { best[k] = best[k - 1]; --k; }
        best[k] = v;
    }
    S.sub_n = 0;
    { uint32_t j = 1; while (j < n_best && best[j++].tot >= best[0].tot - X.PP.max_penalty) ++S.sub_n; }
    S.score2 = best[1].tot; S.score2_m[0] = best[1].s[0]; S.score2_m[1] = best[1].s[1];
    S.tot = best[0].tot; S.dist = best[0].dist; S.mate_score[0] = best[0].s[0]; S.mate_score[1] = best[0].s[1];
    if (best[0].tot < min_score) {
        S.n_alt[0] = S.n_alt[1] = 0;
        if (X.PP.finalize && X.PP.find_orphan && PL.n_chains > 0) {        // orphan recovery (aligner_ksw2.hpp:900-906, 1536-1640): pe_align_kernel's
            PL.status = AF_ST_FALLBACK; S.status = PEF_ST_FALLBACK; G.fb_list[atomicAdd(G.fb_n, 1u)] = (uint32_t)pair; atomicAdd(&G.ctr[AFC_WHY + AF_WHY_LOOP], 1u);
        }
        return;                                                            // not aligned
    }
    if (!X.PP.finalize) { S.status = PEF_ST_ALIGNED; return; }             // learn pass: best_scores[0] is the answer
    const uint32_t fq = best[0].q;
    if (fq == ~0u) return;                                                 // (cannot happen: tot >= min_score belongs to a scored chain)
    S.final_q = fq;
    { const uint32_t cm = PL.cand[2 * fq].strand; S.strand = cm; }          // mate 1 reversed <=> the pair's strand (aligner_ksw2.hpp:2128-2141)
    // the final paired_chain_score: a mate gets its CIGAR when its score reaches its own minimum (chain_score, aligner_ksw2.hpp:2062-2065)
    for (uint32_t k = 0; k < 2 && !fallback; ++k) {
        S.fill_on[k] = best[0].s[k] >= min_m[k] ? 1u : 0u;
        if (!S.fill_on[k]) continue;
        const af_cand_t& C = PL.cand[2 * fq + k];
        const uint32_t t = C.task0;
        for (uint32_t x = 0; x < (uint32_t)C.has_lc + C.has_rc; ++x) {         // an extension must reach the query end for its traceback to start at (mqe_t, qlen - 1)
            const af_res_t R = G.res[t + x];
            const moni_dp_task_t T = G.tasks[t + x];
            const int32_t bound = (T.qlen < T.tlen ? T.qlen : T.tlen) * (int32_t)A.D.sc_mch;
            if (!C.overlap && !(R.mqe + A.D.end_bonus > bound)) { fallback = true; why = AF_WHY_REACH_END; }
        }
        S.ref_pos[k] = cpos[fq][k]; S.as[k] = cs[fq][k];
        const uint32_t n_tb = C.overlap ? 1u : (uint32_t)C.has_lc + C.has_rc + C.n_gap_tasks;
        S.n_tb[k] = n_tb;
        if (!fallback && n_tb) {
            const uint32_t tb0 = atomicAdd(&G.ctr[AFC_TRACED], n_tb);
            if (tb0 + n_tb > G.tb_cap) { fallback = true; why = AF_WHY_CAPACITY; }
            else { S.tb0[k] = tb0; if (C.overlap) G.tb_task[tb0] = C.gtask; else for (uint32_t x = 0; x < n_tb; ++x) G.tb_task[tb0 + x] = t + x; }
        }
    }
    if (fallback) { PL.status = AF_ST_FALLBACK; S.status = PEF_ST_FALLBACK; G.fb_list[atomicAdd(G.fb_n, 1u)] = (uint32_t)pair; atomicAdd(&G.ctr[AFC_WHY + why], 1u); return; }
    S.status = PEF_ST_ALIGNED;
    (void)clen;
}

// ------------------------------------------------------------------------------------------------------------------------------
// pe_finish_kernel: the stitched CIGARs of the final pair (aligner_ksw2.hpp:3049-3108) and the pe_rec_t record the host finishing reads
// ------------------------------------------------------------------------------------------------------------------------------
// the stitched CIGAR of one mate's share: count its operations (out == nullptr) or write them
__device__ __forceinline__ uint32_t pef_stitch(const af_args_t& G, const af_plan_t& PL, const af_cand_t& C, uint32_t tb0, uint32_t* out, bool& ovf) {
    uint32_t n = 0, last = 0;
    bool have = false;
    auto flush = [&]() { if (have) { if (out) out[n] = last; ++n; } };
    auto push = [&](uint32_t op) { flush(); last = op; have = true; };
    auto push_merge_first = [&](uint32_t op, bool first) { if (first && (op & 0xf) == 0 && have && (last & 0xf) == 0) last += op & ~0xfu; else push(op); };
    uint32_t tbx = tb0;
    if (C.overlap) {
        const af_tb_t& T = G.tb[tbx];
        if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) push(T.ops[T.n_ops - 1 - k]);
    } else {
        const af_anchor_t* const AN = PL.an + C.an0;
        if (C.has_lc) { const af_tb_t& T = G.tb[tbx++]; if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) push(T.ops[k]); }
        const uint32_t rc_x = C.has_rc ? tbx++ : 0u;
        for (uint32_t j = 0; j < C.n_an; ++j) {
            const af_anchor_t g = AN[j];
            const uint32_t mlen = g.len;
            if (have && (last & 0xf) == 0) last += mlen << 4; else push(mlen << 4);
            if (j + 1 < C.n_an) {
                if (g.gap_kind == AF_GAP_TASK) { const af_tb_t& T = G.tb[tbx++]; if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) push_merge_first(T.ops[T.n_ops - 1 - k], k == 0); }
                else if (g.gap_kind == AF_GAP_INS) push_merge_first(((uint32_t)(uint16_t)g.gap_val << 4) | 1u, true);
                else if (g.gap_kind == AF_GAP_DEL0) push_merge_first(2u, true);
                else if (g.gap_kind == AF_GAP_1X1) push_merge_first(1u << 4, true);
            }
        }
        if (C.has_rc) { const af_tb_t& T = G.tb[rc_x]; if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) push_merge_first(T.ops[T.n_ops - 1 - k], k == 0); }
    }
    flush();
    return n;
}

__global__ void __launch_bounds__(256) pe_finish_kernel(const pef_args_t X) {
    const af_args_t& G = X.G;
    const ak_args_t& A = G.A;
    const uint64_t r_in = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    const bool active = r_in < A.n_reads;
    pe_sel_t* S = active ? &G.pe_sel[r_in] : nullptr;
    const af_plan_t* PL = active ? &G.plans[r_in] : nullptr;
    const bool mine = active && S->status != PEF_ST_FALLBACK;            // (a pair handed over is written by pe_align_kernel)
    uint32_t nc[2] = {0, 0}, na[2] = {0, 0};
    bool ovf = false;
    if (mine && S->status == PEF_ST_ALIGNED && X.PP.finalize)
        for (uint32_t k = 0; k < 2; ++k) if (S->fill_on[k]) { nc[k] = pef_stitch(G, *PL, PL->cand[2 * S->final_q + k], S->tb0[k], nullptr, ovf); na[k] = S->n_alt[k]; }
    // room in the pools: one atomic per wave and pool
    unsigned long long need_c = nc[0] + nc[1], need_a = na[0] + na[1];
    unsigned long long inc_c = need_c, inc_a = need_a;
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long x = __shfl_up(inc_c, o), y = __shfl_up(inc_a, o); if (lane >= o) { inc_c += x; inc_a += y; } }
    unsigned long long base_c = 0, base_a = 0;
    const unsigned long long tot_c = __shfl(inc_c, 63), tot_a = __shfl(inc_a, 63);
    if (lane == 63) { if (tot_c) base_c = atomicAdd(&X.cursors[0], tot_c); if (tot_a) base_a = atomicAdd(&X.cursors[1], tot_a); }
    base_c = __shfl(base_c, 63); base_a = __shfl(base_a, 63);
    (void)lt_mask;
    if (!mine) return;
    pe_rec_t R;
    R.status = ovf ? 2u : (S->status == PEF_ST_ALIGNED ? 1u : 0u);
    R.strand = S->strand; R.tot = S->tot; R.score2 = S->score2; R.sub_n = S->sub_n; R.pad = 0; R.dist = S->dist;
    R.mate_score[0] = S->mate_score[0]; R.mate_score[1] = S->mate_score[1];
    unsigned long long co = base_c + inc_c - need_c, ao = base_a + inc_a - need_a;
    for (uint32_t k = 0; k < 2; ++k) {
        R.score2_m[k] = S->score2_m[k];
        R.orphan[k] = 0; R.filled[k] = 0; R.ref_pos[k] = 0; R.as[k] = 0; R.n_cigar[k] = 0; R.n_alt[k] = 0; R.cigar_off[k] = 0; R.alt_off[k] = 0;
    }
    if (R.status == 1 && X.PP.finalize) {
        for (uint32_t k = 0; k < 2 && R.status == 1; ++k) {
            if (!S->fill_on[k]) continue;
            if (co + nc[k] > X.cig_cap || ao + na[k] > X.alt_cap) { R.status = 2; break; }
            bool o2 = false;
            pef_stitch(G, *PL, PL->cand[2 * S->final_q + k], S->tb0[k], X.cig_pool + co, o2);
            for (uint32_t i = 0; i < na[k]; ++i) { moni_alt_t x; x.pos = S->alt_pos[k][i]; x.score = S->alt_score[k][i]; x.pad = 0; X.alt_pool[ao + i] = x; }
            R.filled[k] = 1; R.ref_pos[k] = S->ref_pos[k]; R.as[k] = S->as[k];
            R.n_cigar[k] = nc[k]; R.cigar_off[k] = co; R.n_alt[k] = na[k]; R.alt_off[k] = ao;
            co += nc[k]; ao += na[k];
        }
    }
    X.recs[r_in] = R;
}
