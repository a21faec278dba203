// Per-pair logic of the PAIRED-END path, STL-free, on top of align_core.h: compiled for the device (pe_align_kernel: every lane
// runs this code for its own pair, the whole wave runs the DP problems the pairs ask for) and for the host (tests/host_sim replays it).
// Orphan recovery (find_orphan) runs in the same state machine; its local alignment (klib's ksw_align) is a DP_EZ_LOCAL request.
//   aligner_ksw2.hpp:1000-1326  align(paired_alignment_t&, finalize): the four (mate, strand) seed lists, direction filter,
//                               frequency filter, find_chains, get_best_scores, the final paired_chain_score
//   aligner_ksw2.hpp:1329-1431  get_best_scores;  :1471-1534 check_paired_left_MEM;  :2115-2290 paired_chain_score
// What stays on the host for every pair (pe_host.hpp): lift-over of the two CIGARs, MD/NM, MAPQ (SE + PE), flags / TLEN, SAM text.
#pragma once
#include <math.h>

#include "align_core.h"

#ifndef PE_MAX_BEST
#define PE_MAX_BEST 64
#endif

struct pe_params_t {
    ac_params_t P;
    int32_t smismatch, max_penalty;      // max_penalty: aligner_ksw2.hpp:241
    uint32_t filter_dir, finalize;       // finalize == 0: learn_fragment_model's align(al, false)
    double dir_thr;
    float mean, std_dev;                 // paired_alignment_t::mean / std_dev are floats (aligner_ksw2.hpp:684-685)
    uint32_t find_orphan, w;             // w: the separator run between sequences (seqidx::get_w); orphan recovery for the pairs that chain but fail jointly (aligner_ksw2.hpp:900-906)
    double ins_mean, ins_std_dev;        // the model as doubles: the search window of paired_chain_orphan_score (aligner_ksw2.hpp:2398-2420)
    const double* pen_tab;               // the pairing term's penalty for dist < pen_tab_n, computed on the host with the host's libm (null: computed in place)
    uint32_t pen_tab_n;
    uint32_t secondary_chains;           // -Z: find_chains_secondary (aligner_ksw2.hpp:1190-1191)
};

#ifndef DP_EZ_LOCAL
#define DP_EZ_LOCAL 0x100                // a DP request that is klib's ksw_align(KSW_XSTART): result in score / max_t (te) / max_q (qe) / mte (tb) / mte_q (qb)
#endif
enum { PE_O_LOOP = 10, PE_O_WAIT_A, PE_O_WAIT_B, PE_O_FINAL_A, PE_O_FINAL_B };

struct pe_mscore_t { int32_t score; uint32_t pad; uint64_t pos, lft; };                 // score_t
struct pe_pscore_t { int32_t tot; uint32_t paired; long long dist; pe_mscore_t m1, m2; uint64_t chain_i; long long w0, w1; };      // paired_score_t; w0 / w1: orphan_paired_score_t::pos
struct pe_left_t { uint64_t r1, r2, score; };

// Orphan recovery over several waves (pe_kernel.hip: pe_orphan_kernel): the loop over the chains is the expensive part of a pair that fails jointly - two
// DP rounds per chain, a local alignment among them, one chain after the other - and the chains' scores do not depend on one another, only the
// best_scores update that absorbs them does (it merges by region, in chain order).  o_mode 3: the state machine stops where the loop would begin (the
// pair's state stays in its slot); 1: it runs the loop for its block of consecutive chains (block o_part of o_nsplit) and RECORDS each chain's score instead of
// absorbing it; 2: it runs the loop in chain order from the records - no DP - and goes on to the final alignments.  0: the whole thing in one go.
struct pe_orec_t { uint32_t tag, kind; pe_pscore_t sc; };      // kind 1: the chain's score; 2: the chain's requests were beyond the kernel (-> the pair's status 2)
// PE_CSV_COUNT (the host build of pe_big.cpp only): the two numbers of the `-c` line that the state machine alone knows - occurrences of the MEMs the
// direction and frequency filters drop (aligner_ksw2.hpp:1066-1075, 1905-1933), chains check_paired_left_MEM skips (:1354-1358)
#ifdef PE_CSV_COUNT
#define PE_CSV(x) x
#else
#define PE_CSV(x)
#endif
struct pe_ws_t {
    PE_CSV(uint64_t csv_filter; uint64_t csv_skipped;)
    uint32_t o_mode, o_part, o_nsplit, o_tag, o_parked, o_pad;
    pe_orec_t* orec;                     // the pair's records, one per chain
    ac_sec_t sec[AC_MAX_ANCH];           // -Z: the second track of the chaining
    ac_ws_t W;                           // mems, anchors, chains, DP requests, sort stack (W.off / W.m / W.fill / W.best / W.cigar unused)
    uint64_t off[2]; uint32_t m[2];      // the two mates in the resident batch (reads 2p and 2p + 1)
    int32_t min_score_m[2], min_score;
    // selection loop (get_best_scores)
    uint32_t n_best, n_left, n_alt[2];
    int32_t max_m[2];
    pe_pscore_t best[PE_MAX_BEST];
    pe_left_t left[AC_MAX_LEFT];
    uint64_t alt_pos[2][AC_MAX_ALT];
    int32_t alt_score[2][AC_MAX_ALT];
    ac_fill_t fill[2];
    uint32_t fill_on[2];
    // result
    int32_t score2, score2_m[2], sub_n;
    pe_pscore_t final;                   // al.score
    uint32_t strand, filled[2];
    uint32_t orphan[2];                  // the mate was placed by orphan recovery (fill_orphan): no ZS of its own, no alternative hits
    uint32_t o_anch, o_have;             // orphan recovery: the mate that has the chain's anchors; a window was searched
    long long o_start, o_end;            // the window (narrowed once the local alignment is known)
    int32_t o_t_local, o_t_ext, o_t_glob;      // DP requests of the orphan mate
    uint64_t ref_pos[2]; int32_t as[2];
    uint32_t n_cigar[2];
    uint32_t cigar[2][AC_MAX_CIGAR];
};

AC_HD uint64_t pe_dist(uint64_t a, uint64_t b) { return a > b ? a - b : b - a; }

#if defined(__HIP_DEVICE_COMPILE__)
#define PE_DMUL(a, b) __dmul_rn((a), (b))
#define PE_DADD(a, b) __dadd_rn((a), (b))
#else
#define PE_DMUL(a, b) ((a) * (b))
#define PE_DADD(a, b) ((a) + (b))
#endif

// the pairing term (aligner_ksw2.hpp:2176-2181): operations in the reference's order, none contracted.  The penalty depends on dist alone (the
// model and smatch are the call's): the library computes it on the HOST for every dist below pen_tab_n (glibc's erfc / log, what the reference runs
// on) and the kernels look it up; beyond the table (a fragment of more than 8191 bases: the term is far below every score) the device's libm stands in
AC_HD double pe_pair_pen(const pe_params_t& PP, long long dist) {
    double ns = 0.0;
    if (PP.std_dev > 0.0f) ns = (double)(((float)dist - PP.mean) / PP.std_dev);
    return PE_DMUL(PE_DMUL(.721, log(PE_DMUL(2., erfc(PE_DMUL(fabs(ns), M_SQRT1_2))))), (double)PP.P.smatch);
}
AC_HD int32_t pe_pair_total(const pe_params_t& PP, int32_t s1, int32_t s2, long long dist) {
    const int32_t s12 = (int32_t)((uint32_t)s1 + (uint32_t)s2);
    const double pen = (PP.pen_tab && dist >= 0 && (unsigned long long)dist < (unsigned long long)PP.pen_tab_n) ? PP.pen_tab[dist] : pe_pair_pen(PP, dist);
    const double v = PE_DADD(PE_DADD((double)s12, pen), .499);
    int32_t tot = v != v || v <= -2147483648.0 ? INT32_MIN : (v >= 2147483647.0 ? INT32_MIN : (int32_t)v);      // cvttsd2si: out of range -> INT_MIN
    if (tot < 0) tot = 0;
    return tot;
}

// The pair's seeds in the reference's order (aligner_ksw2.hpp:1012-1040 with seed_finder.hpp:311-318): the seeding kernels leave,
// per read, [forward MEMs, reverse-complement MEMs, then for each of them in that order its two halves]; aux marks the halves.
// Returns false if the pair is not chained (or overflowed).
AC_HD_BIG bool pe_init(pe_ws_t& S, const pe_params_t& PP, const moni_mem_t* gm, const uint64_t* rmo, const uint32_t* aux, const uint64_t* occs, uint64_t pair) {
    ac_ws_t& W = S.W;
    const ac_params_t& P = PP.P;
    ac_reset(W);
    PE_CSV(S.csv_filter = 0; S.csv_skipped = 0;)
    S.o_mode = 0; S.o_part = 0; S.o_nsplit = 1; S.o_tag = 0; S.o_parked = 0; S.orec = nullptr;
    S.n_best = S.n_left = 0; S.n_alt[0] = S.n_alt[1] = 0; S.max_m[0] = S.max_m[1] = 0;
    S.score2 = S.score2_m[0] = S.score2_m[1] = 0; S.sub_n = 0; S.strand = 0; S.filled[0] = S.filled[1] = 0; S.n_cigar[0] = S.n_cigar[1] = 0;
    S.orphan[0] = S.orphan[1] = 0; S.o_anch = 0; S.o_have = 0; S.o_start = S.o_end = 0; S.o_t_local = S.o_t_ext = S.o_t_glob = -1; S.final.w0 = S.final.w1 = 0;
    S.final.tot = 0; S.final.paired = 0; S.final.dist = 0; S.final.chain_i = 0;
    S.final.m1.score = S.final.m2.score = 0; S.final.m1.pos = S.final.m2.pos = S.final.m1.lft = S.final.m2.lft = 0;
    const uint64_t r1 = 2 * pair, r2 = 2 * pair + 1;
    // the four find_mems calls: (read, strand bit of the seeding kernels' mate field, mate flags, r_offset)
    struct call_t { uint64_t read; uint32_t rc, mate, roff; };
    call_t calls[4];
    if (PP.filter_dir) {
        calls[0] = {r1, 0u, 0u, 0u}; calls[1] = {r2, 2u, 3u, S.m[0]}; calls[2] = {r2, 0u, 1u, 0u}; calls[3] = {r1, 2u, 2u, S.m[1]};
    } else {
        calls[0] = {r1, 0u, 0u, 0u}; calls[1] = {r1, 2u, 2u, S.m[1]}; calls[2] = {r2, 0u, 1u, 0u}; calls[3] = {r2, 2u, 3u, S.m[0]};
    }
    uint32_t n_dir1 = 0, n_dir2 = 0;
    auto put = [&](const moni_mem_t& g, const call_t& c) -> bool {
        if (W.n_mems >= AC_MAX_MEMS) { W.overflow = 1; return false; }
        ac_mem_t& M = W.mems[W.n_mems++];
        M.pos = g.pos; M.len = g.len; M.idx = g.idx; M.rpos = g.rpos + c.roff; M.mate = c.mate; M.occs = occs + g.occ_off; M.nocc = g.occ_cnt;
        return true;
    };
    for (int k = 0; k < 4; ++k) {                          // the MEMs themselves
        const call_t c = calls[k];
        for (uint64_t g = rmo[c.read]; g < rmo[c.read + 1]; ++g) {
            if (aux[g] >= 0xFFFFFFFDu && aux[g] != 0xFFFFFFFFu) break;          // the halves follow all MEMs of the read
            if (gm[g].mate != c.rc) continue;
            if (!put(gm[g], c)) return false;
            if (k < 2) ++n_dir1; else ++n_dir2;
        }
    }
    for (int k = 0; k < 4; ++k) {                          // populate_seeds: the halves of every long MEM, in the order of the MEMs
        const call_t c = calls[k];
        for (uint64_t g = rmo[c.read]; g < rmo[c.read + 1]; ++g) {
            if (aux[g] >= 0xFFFFFFFDu && aux[g] != 0xFFFFFFFFu) break;
            if (gm[g].mate != c.rc || aux[g] == 0xFFFFFFFFu) continue;
            const uint64_t h = rmo[c.read] + aux[g];
            if (!put(gm[h], c) || !put(gm[h + 1], c)) return false;
        }
    }
    if (PP.filter_dir) {                                   // aligner_ksw2.hpp:1042-1100: only the two plain averages decide
        double a1 = 0.0, a2 = 0.0;
        for (uint32_t i = 0; i < n_dir1; ++i) a1 += (double)W.mems[i].len;
        for (uint32_t i = n_dir1; i < W.n_mems; ++i) a2 += (double)W.mems[i].len;
        if (n_dir1 > 0) a1 = a1 / (double)n_dir1;
        if (n_dir2 > 0) a2 = a2 / (double)n_dir2;
        uint32_t lo = 0, hi = W.n_mems;
        if (a1 > a2 && (a1 - a2) > PP.dir_thr) hi = n_dir1;
        if (a2 > a1 && (a2 - a1) > PP.dir_thr) lo = n_dir1;
        PE_CSV(for (uint32_t i = 0; i < lo; ++i) S.csv_filter += W.mems[i].nocc; for (uint32_t i = hi; i < W.n_mems; ++i) S.csv_filter += W.mems[i].nocc;)
        if (lo > 0) for (uint32_t i = lo; i < hi; ++i) W.mems[i - lo] = W.mems[i];
        W.n_mems = hi - lo;
    }
    if (P.filter_freq) {                                   // seed_freq_filter over what is left
        size_t total = 0;
        for (uint32_t i = 0; i < W.n_mems; ++i) total += W.mems[i].nocc;
        uint32_t k = 0;
        for (uint32_t i = 0; i < W.n_mems; ++i) {
            const double fr = static_cast<double>(W.mems[i].nocc) / total;
            if (fr > P.freq_thr) { PE_CSV(S.csv_filter += W.mems[i].nocc;) continue; }
            if (k != i) W.mems[k] = W.mems[i];
            ++k;
        }
        W.n_mems = k;
    }
    if (W.n_mems == 0) return false;                       // find_chains over no anchors: 0/0 average, no chain
    return ac_chain(W, P, PP.secondary_chains ? S.sec : nullptr);
}

// aligner_ksw2.hpp:1471-1534
AC_HD_BIG bool pe_check_left(pe_ws_t& S, const pe_params_t& PP, uint64_t ci) {
    ac_ws_t& W = S.W;
    const ac_params_t& P = PP.P;
    const ac_chain_t& ch = W.chains[ci];
    uint64_t ref[2] = {0, 0}; bool have[2] = {false, false};
    for (uint32_t k = 0; k < ch.cnt && !(have[0] && have[1]); ++k) {                 // left to right
        const ac_anchor_t& A = W.anch[W.pool[ch.off + ch.cnt - 1 - k]];
        const uint32_t mt = W.mems[A.mem].mate & 1u;
        if (!have[mt]) { have[mt] = true; ref[mt] = ac_seq_off(P, ac_lift(P, ac_occ(W, A.mem, A.occ))) + 1; }
    }
    bool seen = false;
    for (uint32_t k = 0; k < S.n_left; ++k)
        if (pe_dist(S.left[k].r1, ref[0]) < P.region_dist && pe_dist(S.left[k].r2, ref[1]) < P.region_dist && S.left[k].score == (uint64_t)ch.score) seen = true;
    if (seen) return true;
    if (S.n_left >= AC_MAX_LEFT) { W.overflow = 1; return false; }
    S.left[S.n_left].r1 = ref[0]; S.left[S.n_left].r2 = ref[1]; S.left[S.n_left].score = (uint64_t)ch.score; ++S.n_left;
    return false;
}

// check_max_score for one mate (aligner_ksw2.hpp:528-548)
AC_HD void pe_check_max(pe_ws_t& S, int k, int32_t score, uint64_t pos) {
    if (score > S.max_m[k]) { S.max_m[k] = score; S.n_alt[k] = 0; }
    else if (score == S.max_m[k]) {
        if (S.n_alt[k] >= AC_MAX_ALT) { S.W.overflow = 1; return; }
        S.alt_pos[k][S.n_alt[k]] = pos; S.alt_score[k][S.n_alt[k]] = score; ++S.n_alt[k];
    }
}

// a scored chain comes back into get_best_scores (aligner_ksw2.hpp:1368-1400)
AC_HD_BIG void pe_absorb_best(pe_ws_t& S, const pe_params_t& PP, const pe_pscore_t& sc);
AC_HD_BIG void pe_absorb(pe_ws_t& S, const pe_params_t& PP, const pe_pscore_t& sc) {
    pe_check_max(S, 0, sc.m1.score, sc.m1.pos);
    pe_check_max(S, 1, sc.m2.score, sc.m2.pos);
    if (S.W.overflow) return;
    pe_absorb_best(S, PP, sc);
}
// the best_scores update shared by get_best_scores and orphan_recovery (aligner_ksw2.hpp:1376-1400, 1572-1596)
AC_HD_BIG void pe_absorb_best(pe_ws_t& S, const pe_params_t& PP, const pe_pscore_t& sc) {
    const ac_params_t& P = PP.P;
    if (sc.tot >= S.min_score) {
        bool replaced = false;
        pe_pscore_t zero;
        zero.tot = 0; zero.paired = 0; zero.dist = 0; zero.chain_i = S.W.i;
        zero.m1.score = zero.m2.score = 0; zero.m1.pad = zero.m2.pad = 0; zero.m1.pos = zero.m2.pos = zero.m1.lft = zero.m2.lft = 0; zero.w0 = zero.w1 = 0;
        for (uint32_t j = 0; j < S.n_best; ++j) {
            if (pe_dist(S.best[j].m1.lft, sc.m1.lft) < P.region_dist && pe_dist(S.best[j].m2.lft, sc.m2.lft) < P.region_dist) {
                if (sc.tot > S.best[j].tot) {
                    if (replaced) S.best[j] = zero;
                    else { S.best[j] = sc; replaced = true; }
                } else { j = S.n_best; replaced = true; }
            }
        }
        if (!replaced) {
            if (S.n_best >= PE_MAX_BEST) { S.W.overflow = 1; return; }
            S.best[S.n_best++] = sc;
        }
    }
}

// the anchors of chain ci split by mate, left to right (aligner_ksw2.hpp:2149-2162); false: a mate's share does not fit
AC_HD_BIG bool pe_split_chain(pe_ws_t& S, uint64_t ci) {
    ac_ws_t& W = S.W;
    const ac_chain_t& ch = W.chains[ci];
    S.fill[0].n_an = S.fill[1].n_an = 0;
    for (uint32_t k = 0; k < ch.cnt; ++k) {
        const ac_anchor_t& A = W.anch[W.pool[ch.off + ch.cnt - 1 - k]];
        ac_fill_t& F = S.fill[W.mems[A.mem].mate & 1u];
        if (F.n_an >= AC_MAX_FILL) { W.overflow = 1; return false; }
        F.an_mem[F.n_an] = A.mem; F.an_occ[F.n_an] = A.occ; ++F.n_an;
    }
    const uint32_t cm = ch.mate;
    S.strand = (cm == 0 || cm == 3) ? 0u : 1u;           // aligner_ksw2.hpp:2128-2141
    return true;
}

// queue the DP problems of the fills that are switched on
AC_HD_BIG bool pe_fills_begin(pe_ws_t& S, const pe_params_t& PP, bool score_only) {
    ac_ws_t& W = S.W;
    W.n_tasks = 0;
    for (int k = 0; k < 2; ++k) {
        if (!S.fill_on[k]) continue;
        if (W.n_tasks + S.fill[k].n_an + 1 > AC_MAX_TASKS) { W.overflow = 1; return false; }
        if (!ac_fill_begin_g(W, PP.P, S.fill[k], S.off[k], S.m[k], score_only)) return false;
    }
    return true;
}

// ---- orphan recovery (aligner_ksw2.hpp:1536-1640 orphan_recovery, :2329-2560 paired_chain_orphan_score, :2566-2720 fill_orphan) ----
// the whole mate k in the orientation the pair's strand gives it (mate1 / mate2_rev, or mate1_rev / mate2), as a DP query
AC_HD void pe_whole_mate(const pe_ws_t& S, int k, uint64_t& q_off, int& qmode) {
    const bool rev = k == 0 ? S.strand != 0 : S.strand == 0;
    if (!rev) { q_off = S.off[k]; qmode = 0; } else { q_off = S.off[k] + S.m[k] - 1; qmode = DP_Q_REV | DP_Q_COMP; }
}
// chain ci: the anchored mate's score-only fill and the local search of the other mate in the window the model predicts
AC_HD_BIG bool pe_orphan_begin(pe_ws_t& S, const pe_params_t& PP, uint64_t ci) {
    ac_ws_t& W = S.W;
    const ac_params_t& P = PP.P;
    if (!pe_split_chain(S, ci)) return false;
    const ac_chain_t& ch = W.chains[ci];
    uint64_t lm = ~0ull, rm = 0;
    for (uint32_t k = 0; k < ch.cnt; ++k) {
        const ac_anchor_t& A = W.anch[W.pool[ch.off + k]];
        const uint64_t o = ac_occ(W, A.mem, A.occ);
        if (o + W.mems[A.mem].len > rm) rm = o + W.mems[A.mem].len;
        if (o < lm) lm = o;
    }
    S.o_anch = S.fill[0].n_an > 0 ? 0u : 1u;
    S.fill_on[S.o_anch] = 1; S.fill_on[1 - S.o_anch] = 0;
    if (!pe_fills_begin(S, PP, true)) return false;
    const long long lim = (long long)(P.n_text - PP.w);
    long long start, end;
    if (S.o_anch == 0) { start = (long long)(rm + (uint64_t)(long long)floor(PP.ins_mean - 4 * PP.ins_std_dev)); end = (long long)(rm + (uint64_t)(long long)ceil(PP.ins_mean + 4 * PP.ins_std_dev)); }
    else { start = (long long)(lm + (uint64_t)(long long)floor(-PP.ins_mean - 4 * PP.ins_std_dev)); end = (long long)(lm + (uint64_t)(long long)ceil(-PP.ins_mean + 4 * PP.ins_std_dev)); }
    if (start < 0) start = 0;
    if (start > lim) start = lim;
    if (end > lim) end = lim;
    S.o_start = start; S.o_end = end; S.o_have = start < end ? 1u : 0u;
    S.o_t_local = S.o_t_ext = S.o_t_glob = -1;
    if (S.o_have) {
        if (W.n_tasks >= AC_MAX_TASKS) { W.overflow = 1; return false; }
        uint64_t q_off; int qmode;
        const int k = 1 - (int)S.o_anch;
        pe_whole_mate(S, k, q_off, qmode);
        moni_dp_task_t& t = W.tasks[W.n_tasks];
        t.q_off = q_off; t.t_off = (uint64_t)start; t.qlen = (int)S.m[k]; t.tlen = (int)(end - start + 1); t.flag = DP_EZ_LOCAL; t.reserved = DP_Q_READS | DP_T_TEXT | qmode;
        S.o_t_local = (int32_t)W.n_tasks++;
    }
    return true;
}
// the orphan loop: every chain in turn, then the best of them (returns true if DP requests were queued)
AC_HD_BIG bool pe_orphan_advance(pe_ws_t& S, const pe_params_t& PP) {
    ac_ws_t& W = S.W;
    while (W.stage == PE_O_LOOP && !W.overflow) {
        if (W.i < W.n_chains) {
            // (blocks of consecutive chains, not a stride: chains of equal score - the copies of one locus on the haplotypes - are neighbours, and a wave answers a
            // chain's DP requests from its memo when the previous chain asked the same of the same sequence)
            if (S.o_mode == 1 && (uint32_t)(W.i / ((W.n_chains + S.o_nsplit - 1) / S.o_nsplit)) != S.o_part) { ++W.i; continue; }          // another wave's chain
            if (S.o_mode == 2 && S.orec[W.i].tag == S.o_tag) {                                            // scored by the waves of the pass before
                const pe_orec_t& R = S.orec[W.i];
                if (R.kind != 1) { W.overflow = 1; return false; }
                pe_absorb_best(S, PP, R.sc);
                ++W.i;
                continue;
            }
            if (!pe_orphan_begin(S, PP, W.i)) return false;
            W.stage = PE_O_WAIT_A;
            return true;
        }
        if (S.o_mode == 1) { W.stage = AC_DONE; return false; }                                           // this wave's chains are scored
        while (S.n_best < 2) {
            pe_pscore_t& z = S.best[S.n_best++];
            z.tot = 0; z.paired = 0; z.dist = 0; z.chain_i = W.n_chains;
            z.m1.score = z.m2.score = 0; z.m1.pad = z.m2.pad = 0; z.m1.pos = z.m2.pos = z.m1.lft = z.m2.lft = 0; z.w0 = z.w1 = 0;
        }
        lsort::sort(S.best, (long)S.n_best, [](const pe_pscore_t& x, const pe_pscore_t& y) {            // orphan_paired_score_t::operator>
            return x.tot > y.tot || (x.tot == y.tot && x.m1.lft > y.m1.lft) || (x.tot == y.tot && x.m1.lft == y.m1.lft && x.m2.lft > y.m2.lft);
        }, W.sort_stack);
        if (S.best[0].tot < S.min_score) { W.stage = AC_DONE; return false; }
        S.sub_n = 0;
        { uint32_t j = 1; while (j < S.n_best && S.best[j++].tot >= S.best[0].tot - PP.max_penalty) ++S.sub_n; }
        S.score2 = S.best[1].tot; S.score2_m[0] = S.best[1].m1.score; S.score2_m[1] = S.best[1].m2.score;
        S.final = S.best[0];
        // the final paired_chain_orphan_score: the anchored mate's chain_score with its CIGAR, the other mate aligned globally over the window
        if (S.best[0].chain_i >= W.n_chains) { W.stage = AC_DONE; return false; }
        if (!pe_split_chain(S, S.best[0].chain_i)) return false;
        S.o_anch = S.fill[0].n_an > 0 ? 0u : 1u;
        const int32_t a_score = S.o_anch == 0 ? S.best[0].m1.score : S.best[0].m2.score;
        S.fill_on[S.o_anch] = a_score >= S.min_score_m[S.o_anch]; S.fill_on[1 - S.o_anch] = 0;
        if (!pe_fills_begin(S, PP, false)) return false;
        S.o_start = S.best[0].w0; S.o_end = S.best[0].w1; S.o_have = S.o_start < S.o_end ? 1u : 0u;
        S.o_t_local = S.o_t_ext = S.o_t_glob = -1;
        if (S.o_have) {
            if (W.n_tasks >= AC_MAX_TASKS) { W.overflow = 1; return false; }
            uint64_t q_off; int qmode;
            const int k = 1 - (int)S.o_anch;
            pe_whole_mate(S, k, q_off, qmode);
            moni_dp_task_t& t = W.tasks[W.n_tasks];
            t.q_off = q_off; t.t_off = (uint64_t)S.o_start; t.qlen = (int)S.m[k]; t.tlen = (int)(S.o_end - S.o_start + 1); t.flag = DP_EZ_RIGHT; t.reserved = DP_Q_READS | DP_T_TEXT | qmode;
            S.o_t_glob = (int32_t)W.n_tasks++;
        }
        W.stage = PE_O_FINAL_A;
        return true;
    }
    return false;
}

// Runs get_best_scores until the pair needs DP results or is done.  Returns true if fills were started.
AC_HD_BIG bool pe_advance(pe_ws_t& S, const pe_params_t& PP) {
    ac_ws_t& W = S.W;
    const ac_params_t& P = PP.P;
    while (W.stage == AC_LOOP && !W.overflow) {
        if (W.i < W.n_chains && W.n_diff < P.check_k) {
            { const uint64_t v = (uint64_t)W.chains[W.i].score; bool f = false; for (uint32_t q = 0; q < W.n_diff; ++q) f = f || W.diff[q] == v; if (!f) W.diff[W.n_diff++] = v; }
            if (P.left_mem_check && pe_check_left(S, PP, W.i)) { ++W.i; PE_CSV(++S.csv_skipped;) continue; }
            if (W.overflow) return false;
            if (W.n_diff < P.check_k) {
                if (!W.chains[W.i].paired) {              // paired_chain_score returns the empty score (aligner_ksw2.hpp:2145)
                    pe_pscore_t z;
                    z.tot = 0; z.paired = 0; z.dist = 0; z.chain_i = W.i;
                    z.m1.score = z.m2.score = 0; z.m1.pad = z.m2.pad = 0; z.m1.pos = z.m2.pos = z.m1.lft = z.m2.lft = 0; z.w0 = z.w1 = 0;
                    pe_absorb(S, PP, z);
                    ++W.i;
                    continue;
                }
                if (!pe_split_chain(S, W.i)) return false;
                S.fill_on[0] = S.fill_on[1] = 1;
                if (!pe_fills_begin(S, PP, true)) return false;
                W.stage = AC_WAIT_A;
                return true;
            }
            continue;
        }
        // after the loop (aligner_ksw2.hpp:1402-1431)
        while (S.n_best < 2) {
            pe_pscore_t& z = S.best[S.n_best++];
            z.tot = 0; z.paired = 0; z.dist = 0; z.chain_i = W.n_chains;
            z.m1.score = z.m2.score = 0; z.m1.pad = z.m2.pad = 0; z.m1.pos = z.m2.pos = z.m1.lft = z.m2.lft = 0; z.w0 = z.w1 = 0;
        }
        lsort::sort(S.best, (long)S.n_best, [](const pe_pscore_t& x, const pe_pscore_t& y) {            // paired_score_t::operator>
            return x.tot > y.tot || (x.tot == y.tot && x.m1.lft > y.m1.lft) || (x.tot == y.tot && x.m1.lft == y.m1.lft && x.m2.lft > y.m2.lft);
        }, W.sort_stack);
        S.sub_n = 0;
        { uint32_t j = 1; while (j < S.n_best && S.best[j++].tot >= S.best[0].tot - PP.max_penalty) ++S.sub_n; }
        S.score2 = S.best[1].tot; S.score2_m[0] = S.best[1].m1.score; S.score2_m[1] = S.best[1].m2.score;
        S.final = S.best[0];
        if (S.best[0].tot < S.min_score) {
            S.n_alt[0] = S.n_alt[1] = 0;
            if (PP.finalize && PP.find_orphan && W.n_chains > 0) {
                S.n_best = 0; W.i = 0; W.stage = PE_O_LOOP;
                if (S.o_mode == 3) { S.o_parked = 1; W.n_tasks = 0; return false; }          // the loop is other launches' (pe_orphan_kernel)
                return pe_orphan_advance(S, PP);
            }      // aligner_ksw2.hpp:900-906
            W.stage = AC_DONE;
            return false;
        }
        if (!PP.finalize) { W.aligned = 1; W.stage = AC_DONE; return false; }                        // learn pass: best_scores[0] is the answer
        if (S.best[0].chain_i >= W.n_chains || !W.chains[S.best[0].chain_i].paired) { W.stage = AC_DONE; return false; }      // (cannot happen: tot >= min_score)
        if (!pe_split_chain(S, S.best[0].chain_i)) return false;
        S.fill_on[0] = S.best[0].m1.score >= S.min_score_m[0]; S.fill_on[1] = S.best[0].m2.score >= S.min_score_m[1];
        if (!pe_fills_begin(S, PP, false)) return false;
        W.stage = AC_FINAL_WAIT_A;
        return true;
    }
    return false;
}

// Drive the pair: consume the results of the DP problems it queued last and continue until it queues new ones (W.n_tasks > 0),
// finishes (W.stage == AC_DONE) or overflows.
AC_HD_BIG void pe_drive(pe_ws_t& S, const pe_params_t& PP, const moni_dp_result_t* res, const uint32_t* cig) {
    ac_ws_t& W = S.W;
    const ac_params_t& P = PP.P;
    while (!W.overflow) {
        switch (W.stage) {
            case AC_LOOP:
                W.n_tasks = 0;
                if (!pe_advance(S, PP)) return;
                if (W.n_tasks > 0) return;
                break;
            case AC_WAIT_A:
            case AC_WAIT_B: {
                if (W.stage == AC_WAIT_A) {
                    bool more = false;
                    W.n_tasks = 0;
                    for (int k = 0; k < 2; ++k) if (ac_fill_after_ext_g(W, P, S.fill[k], S.off[k], S.m[k], res)) more = true;
                    if (more) { W.stage = AC_WAIT_B; return; }
                } else {
                    for (int k = 0; k < 2; ++k) if (S.fill[k].t_glob >= 0) S.fill[k].score = res[S.fill[k].t_glob].score;
                }
                pe_pscore_t sc;
                sc.chain_i = W.i; sc.paired = 1; sc.w0 = sc.w1 = 0;
                for (int k = 0; k < 2; ++k) {
                    ac_fill_t& F = S.fill[k];
                    if (!ac_valid(P, F.ref_pos, F.ref_len)) F.score = INT32_MIN;
                    pe_mscore_t& ms = k ? sc.m2 : sc.m1;
                    ms.score = F.score; ms.pad = 0; ms.pos = F.score_pos; ms.lft = ac_lift(P, F.score_pos);
                }
                sc.dist = (long long)pe_dist(sc.m2.pos, sc.m1.pos + (uint64_t)S.m[0]);
                sc.tot = pe_pair_total(PP, sc.m1.score, sc.m2.score, sc.dist);
                pe_absorb(S, PP, sc);
                ++W.i;
                W.stage = AC_LOOP;
                break;
            }
            case AC_FINAL_WAIT_A:
            case AC_FINAL_WAIT_B: {
                // a fill without overlapping MEMs is stitched from this round's results; one with overlap waits for its global problem
                const bool first = W.stage == AC_FINAL_WAIT_A;
                bool more = false;
                if (first) W.n_tasks = 0;
                for (int k = 0; k < 2; ++k) {
                    if (!S.fill_on[k]) continue;
                    ac_fill_t& F = S.fill[k];
                    if (first && ac_fill_after_ext_g(W, P, F, S.off[k], S.m[k], res)) { more = true; continue; }
                    if (!ac_fill_final_g(W, P, F, res, cig, S.cigar[k], S.n_cigar[k])) return;
                    S.filled[k] = 1; S.ref_pos[k] = F.ref_pos; S.as[k] = F.score;
                    S.fill_on[k] = 0;
                }
                if (more) { W.stage = AC_FINAL_WAIT_B; return; }
                W.aligned = 1; W.stage = AC_DONE; W.n_tasks = 0;
                return;
            }
            case PE_O_LOOP:
                W.n_tasks = 0;
                if (!pe_orphan_advance(S, PP)) return;
                if (W.n_tasks > 0) return;
                break;
            case PE_O_WAIT_A:
            case PE_O_WAIT_B: {
                const int a = (int)S.o_anch, o = 1 - a;
                const bool second = W.stage == PE_O_WAIT_B;
                ac_fill_t& FA = S.fill[a];
                if (!second) {
                    int te = -1, tb = -1;
                    if (S.o_t_local >= 0) { te = res[S.o_t_local].max_t; tb = res[S.o_t_local].mte; }
                    W.n_tasks = 0;
                    bool more = ac_fill_after_ext_g(W, P, FA, S.off[a], S.m[a], res);
                    S.o_t_ext = -1;
                    if (S.o_have) {                                   // fill_orphan: end = start + r.te; start += r.tb; then the extension score over the narrowed window
                        S.o_end = S.o_start + te; S.o_start += tb;
                        if (tb >= 0 && te >= tb) {
                            if (W.n_tasks >= AC_MAX_TASKS) { W.overflow = 1; return; }
                            uint64_t q_off; int qmode;
                            pe_whole_mate(S, o, q_off, qmode);
                            moni_dp_task_t& t = W.tasks[W.n_tasks];
                            t.q_off = q_off; t.t_off = (uint64_t)S.o_start; t.qlen = (int)S.m[o]; t.tlen = te - tb + 1; t.flag = DP_EZ_SCORE_ONLY; t.reserved = DP_Q_READS | DP_T_TEXT | qmode;
                            S.o_t_ext = (int32_t)W.n_tasks++;
                            more = true;
                        }
                    }
                    if (more) { W.stage = PE_O_WAIT_B; return; }
                } else if (FA.t_glob >= 0) FA.score = res[FA.t_glob].score;
                pe_pscore_t sc;
                sc.chain_i = W.i; sc.paired = 1; sc.w0 = S.o_start; sc.w1 = S.o_end;
                pe_mscore_t ms[2];
                if (!ac_valid(P, FA.ref_pos, FA.ref_len)) FA.score = INT32_MIN;
                ms[a].score = FA.score; ms[a].pad = 0; ms[a].pos = FA.score_pos;
                ms[o].score = 0; ms[o].pad = 0; ms[o].pos = 0;
                if (S.o_have) {
                    ms[o].score = (second && S.o_t_ext >= 0) ? res[S.o_t_ext].score : -0x40000000;      // KSW_NEG_INF: no extension ran
                    ms[o].pos = (uint64_t)S.o_start;
                    if (!ac_valid(P, (uint64_t)S.o_start, (uint64_t)(S.o_end - S.o_start + 1))) ms[o].score = INT32_MIN;
                }
                ms[0].lft = ac_lift(P, ms[0].pos); ms[1].lft = ac_lift(P, ms[1].pos);
                sc.m1 = ms[0]; sc.m2 = ms[1];
                sc.dist = (long long)pe_dist(sc.m2.pos, sc.m1.pos + (uint64_t)S.m[0]);
                sc.tot = pe_pair_total(PP, sc.m1.score, sc.m2.score, sc.dist);
                if (S.o_mode == 1) { pe_orec_t& R = S.orec[W.i]; R.sc = sc; R.kind = 1; R.tag = S.o_tag; }
                else pe_absorb_best(S, PP, sc);
                ++W.i;
                W.stage = PE_O_LOOP;
                break;
            }
            case PE_O_FINAL_A:
            case PE_O_FINAL_B: {
                const int a = (int)S.o_anch, o = 1 - a;
                const bool first = W.stage == PE_O_FINAL_A;
                ac_fill_t& FA = S.fill[a];
                if (first) {
                    if (S.o_t_glob >= 0) {                            // fill_orphan(.., false, sam): the whole mate against the window, with its CIGAR
                        const moni_dp_result_t& r = res[S.o_t_glob];
                        if (r.n_cigar > AC_MAX_CIGAR) { W.overflow = 1; return; }
                        for (uint32_t k = 0; k < r.n_cigar; ++k) S.cigar[o][k] = cig[r.cigar_off + k];
                        S.n_cigar[o] = r.n_cigar; S.filled[o] = 1; S.orphan[o] = 1; S.ref_pos[o] = (uint64_t)S.o_start; S.as[o] = r.score;
                    }
                    W.n_tasks = 0;
                    if (S.fill_on[a] && ac_fill_after_ext_g(W, P, FA, S.off[a], S.m[a], res)) { W.stage = PE_O_FINAL_B; return; }
                }
                if (S.fill_on[a]) {
                    if (!ac_fill_final_g(W, P, FA, res, cig, S.cigar[a], S.n_cigar[a])) return;
                    S.filled[a] = 1; S.ref_pos[a] = FA.ref_pos; S.as[a] = FA.score;
                }
                pe_mscore_t ms[2];
                ms[a] = a == 0 ? S.final.m1 : S.final.m2;            // chain_score's score-only values
                ms[o].score = 0; ms[o].pad = 0; ms[o].pos = 0;
                if (S.filled[o]) {                                   // score / pos only when the lifted alignment spans reference bases (aligner_ksw2.hpp:2690-2705)
                    const uint32_t sid = ac_seq_of(P, S.ref_pos[o]);
                    const moni_lift_seq_t Lq = P.lift_seqs[sid];
                    const int nl = lift_cigar(P.lift_runs + Lq.run_off, Lq.n_runs, S.ref_pos[o] - Lq.start, S.cigar[o], S.n_cigar[o], W.cigar, AC_MAX_CIGAR);
                    if (nl < 0) { W.overflow = 1; return; }
                    uint64_t l_len = 0;
                    for (int k = 0; k < nl; ++k) { const int op = W.cigar[k] & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) l_len += W.cigar[k] >> 4; }
                    if (l_len > 0) { ms[o].score = S.as[o]; ms[o].pos = S.ref_pos[o]; }
                }
                ms[0].lft = ac_lift(P, ms[0].pos); ms[1].lft = ac_lift(P, ms[1].pos);
                S.final.m1 = ms[0]; S.final.m2 = ms[1];
                S.final.dist = (long long)pe_dist(ms[1].pos, ms[0].pos + (uint64_t)S.m[0]);
                S.final.tot = pe_pair_total(PP, ms[0].score, ms[1].score, S.final.dist);
                W.aligned = 1; W.stage = AC_DONE; W.n_tasks = 0;
                return;
            }
            default:
                W.n_tasks = 0;
                return;
        }
    }
}
