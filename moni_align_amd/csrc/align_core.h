// Per-read logic of the full single-end path, STL-free, compiled for the device (align_kernel: every lane runs this code
// for its own read, the whole wave runs the DP problems the reads ask for) and for the host (tests/host_sim replays it).  It is the
// same algorithm as align_host.hpp — frequency filter, chaining, the chain-selection loop, fill_chain, CIGAR stitching —
// over fixed-capacity arrays; a read that does not fit the capacities is flagged and goes through the host pipeline.
//   chain.hpp:221-438, aligner_ksw2.hpp:328-521, 528-597, 1905-1933, 2018-2098, 2752-3108
// What stays on the host for every read: MAPQ (double log), names, SAM text (MD/NM are computed in align_kernel).
#pragma once
#include <stdint.h>

#include "../../include/moni_hip.h"
#include "sort_emul.h"
#include "lift_core.h"

#if defined(__HIPCC__)
#define AC_HD __host__ __device__ __forceinline__
#define AC_HD_BIG __host__ __device__ __attribute__((noinline))     // real calls on the device: the kernel stays compilable
#else
#define AC_HD inline
#define AC_HD_BIG inline
#endif

#ifndef DP_EZ_SCORE_ONLY
#define DP_EZ_SCORE_ONLY 0x01
#define DP_EZ_RIGHT 0x02
#define DP_EZ_EXTZ_ONLY 0x40
#define DP_Q_READS 0x01
#define DP_Q_REV 0x02
#define DP_Q_COMP 0x04
#define DP_T_TEXT 0x08
#define DP_T_REV 0x10
#endif

#ifndef AC_MAX_MEMS                // (pe_big.cpp compiles this header a second time, in its own namespace, with large capacities)
#define AC_MAX_MEMS 96
#define AC_MAX_ANCH 512
#define AC_MAX_CHAINS 256
#define AC_MAX_POOL 1024        // anchors of all chains
#define AC_MAX_BEST 64
#define AC_MAX_LEFT 256
#define AC_MAX_ALT 64
#define AC_MAX_FILL 16          // anchors of one chain that fill_chain handles
#define AC_MAX_CIGAR 512
#endif
#define AC_MAX_TASKS (2 * (AC_MAX_FILL + 2))      // one round of a pair: both mates' fills (pe_core.h)

struct ac_mem_t { uint64_t pos; const uint64_t* occs; uint32_t len, idx, rpos, mate, nocc; };
struct ac_anchor_t { uint64_t x; uint32_t mem, occ; };                 // x = reference end of the anchor (the sort key)
struct ac_node_t { uint64_t x; uint16_t rpos, len; uint8_t mate, pad0; uint16_t pad1; int32_t f, p, t, msc; };      // 32 bytes
struct ac_chain_t { long long score; uint32_t mate, off, cnt, paired; };      // anchors (right to left) at pool[off .. off+cnt); paired: chain.hpp:186
struct ac_start_t { long long f; uint64_t j; };
struct ac_best_t { int32_t score; uint64_t lft; uint64_t idx; };
struct ac_left_t { uint64_t ref; uint64_t score; };

struct ac_params_t {
    uint32_t min_len, ext_len, check_k, region_dist, filter_freq, left_mem_check;
    double freq_thr;
    int32_t smatch, gapo, gapo2, gape, gape2;
    long long max_dist_x, max_dist_y, max_iter, max_pred, min_chain_score, min_chain_length;
    uint64_t n_text;
    uint32_t n_seq;
    const uint64_t* seq_starts;      // n_seq + 1 onsets
    const moni_lift_seq_t* lift_seqs; // one lift per sequence (lift_core.h)
    const moni_lift_run_t* lift_runs;
    const uint64_t* pdir;            // position directory (lift_core.h) or nullptr (binary searches)
};

enum { AC_LOOP = 0, AC_WAIT_A, AC_WAIT_B, AC_FINAL_WAIT_A, AC_FINAL_WAIT_B, AC_DONE };

struct ac_fill_t {
    uint32_t score_only, n_an, strand, overlap;
    uint32_t an_mem[AC_MAX_FILL], an_occ[AC_MAX_FILL];
    uint64_t lcs_len, rcs_len, rcs_occ, ref_pos, ref_len;
    int32_t t_lc, t_rc, t_glob, lc_mqe_t, rc_mqe_t;
    int32_t t_gap[AC_MAX_FILL], gap_score[AC_MAX_FILL];
    uint32_t gap_cig[AC_MAX_FILL];
    int32_t score;
    uint64_t score_pos;
};

#if defined(__HIP_DEVICE_COMPILE__)
#define AC_CLOCK() ((unsigned long long)clock64())
#else
#define AC_CLOCK() 0ull
#endif

struct ac_ws_t {
    lsort::frame sort_stack[lsort::STACK_FRAMES];
    unsigned long long prof[6];      // device only: cycles in ac_init's parts (load, sort, chain DP, backtrack) 
    // read
    uint64_t off; uint32_t m; int32_t min_score;
    // seeds after the frequency filter, anchors, chaining scratch, chains
    uint32_t n_mems, n_anch, n_chains, pool_used;
    ac_mem_t mems[AC_MAX_MEMS];
    ac_anchor_t anch[AC_MAX_ANCH];
    ac_node_t node[AC_MAX_ANCH];
    ac_start_t starts[AC_MAX_CHAINS];
    ac_chain_t chains[AC_MAX_CHAINS];
    uint32_t pool[AC_MAX_POOL];
    int32_t score_cache[AC_MAX_CHAINS];
    // selection loop
    uint64_t i;
    uint32_t n_diff, n_best, n_left, n_alt;
    uint64_t diff[8];
    ac_best_t best[AC_MAX_BEST];
    ac_left_t left[AC_MAX_LEFT];
    uint64_t alt_pos[AC_MAX_ALT];
    int32_t alt_score[AC_MAX_ALT];
    int32_t max_score, score2;
    uint64_t final_chain;
    uint32_t stage, aligned, overflow;
    ac_fill_t fill;
    // DP request of the current round
    uint32_t n_tasks;
    moni_dp_task_t tasks[AC_MAX_TASKS];
    // final alignment
    uint32_t n_cigar;
    uint32_t cigar[AC_MAX_CIGAR];
};

AC_HD int ac_ilog2(uint32_t v) { int r = 0; while (v >>= 1) ++r; return r; }

AC_HD uint64_t ac_rank1(const ac_params_t& P, uint64_t i) {           // number of onsets < i
    uint32_t lo = 0, hi = P.n_seq + 1;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (P.seq_starts[mid] < i) lo = mid + 1; else hi = mid; }
    return lo;
}
// the sequence a text position lies in (seqidx: rank1(pos + 1) - 1), and with run_hint the lift run at or before its haplotype position
AC_HD uint32_t ac_seq_of(const ac_params_t& P, uint64_t pos, uint32_t* run_hint = nullptr) {
    if (!P.pdir) { if (run_hint) *run_hint = 0xFFFFFFFFu; return (uint32_t)(ac_rank1(P, pos + 1) - 1); }
    uint64_t b = pos >> MONI_PDIR_SHIFT;
    const uint64_t nb = (P.n_text >> MONI_PDIR_SHIFT) + 1;
    if (b > nb) b = nb;
    const uint64_t e = P.pdir[b];
    uint32_t sid = (uint32_t)e;
    bool moved = false;
    while (sid + 1 < P.n_seq && pos >= P.lift_seqs[sid + 1].start) { ++sid; moved = true; }
    if (run_hint) *run_hint = moved ? 0xFFFFFFFFu : (uint32_t)(e >> 32);
    return sid;
}
AC_HD uint64_t ac_seq_off(const ac_params_t& P, uint64_t pos) {          // index(pos).second
    if (!P.pdir) { const uint64_t rk = ac_rank1(P, pos + 1); return pos - P.seq_starts[rk - 1]; }
    return pos - P.lift_seqs[ac_seq_of(P, pos)].start;
}
AC_HD bool ac_valid(const ac_params_t& P, uint64_t pos, uint64_t len) {
    if (!P.pdir) { const uint64_t rk = ac_rank1(P, pos + 1); return pos + len <= P.seq_starts[rk]; }
    return pos + len <= P.lift_seqs[ac_seq_of(P, pos)].end;
}

// liftidx::lift (liftidx.hpp:89-95)
AC_HD uint64_t ac_lift(const ac_params_t& P, uint64_t pos) {
    uint32_t hint;
    const uint32_t sid = ac_seq_of(P, pos, &hint);
    const moni_lift_seq_t L = P.lift_seqs[sid];
    const uint64_t start = pos - L.start;
    if (hint == 0xFFFFFFFFu) return L.second + lift_pos(P.lift_runs + L.run_off, L.n_runs, start);
    // the directory's run holds a haplotype position <= start: walk forward to the last such run (lift_find)
    uint32_t k = hint;
    const uint32_t last = L.run_off + L.n_runs;
    moni_lift_run_t R = P.lift_runs[k];
    while (k + 1 < last) { const moni_lift_run_t N = P.lift_runs[k + 1]; if ((uint64_t)N.hap > start) break; R = N; ++k; }
    const uint64_t x = (uint64_t)R.col + (start - (uint64_t)R.hap);
    return L.second + (uint64_t)R.ref + ((R.flags & MONI_LIFT_INS) ? 0ull : x - (uint64_t)R.col);
}

AC_HD uint64_t ac_occ(const ac_ws_t& W, uint32_t mem, uint32_t occ) { return W.mems[mem].occs[occ]; }

// ---- seeds -> filtered mems -> anchors -> chains (aligner_ksw2.hpp:342,382; chain.hpp:221-438) ----
// returns false if the read is not chained (or overflowed: W.overflow)
AC_HD void ac_reset(ac_ws_t& W) {
    W.stage = AC_DONE; W.aligned = 0; W.overflow = 0; W.n_cigar = 0; W.n_tasks = 0;
    W.i = 0; W.n_diff = W.n_best = W.n_left = W.n_alt = 0; W.max_score = 0; W.score2 = 0; W.final_chain = 0;
    W.n_mems = W.n_anch = W.n_chains = W.pool_used = 0;
}
struct ac_sec_t { int32_t f, p, t, msc; };          // the second track of find_chains_secondary (-Z, paired-end): f_sec / p_sec / t_sec / msc_sec of an anchor
AC_HD_BIG bool ac_chain(ac_ws_t& W, const ac_params_t& P, ac_sec_t* sec = nullptr);
AC_HD_BIG bool ac_init(ac_ws_t& W, const ac_params_t& P, const moni_mem_t* gm, uint64_t a, uint64_t b, const uint64_t* occs) {
    ac_reset(W);
    unsigned long long pc0 = AC_CLOCK();
    size_t total = 0;
    for (uint64_t k = a; k < b; ++k) total += gm[k].occ_cnt;
    for (uint64_t k = a; k < b; ++k) {
        const moni_mem_t& g = gm[k];
        if (P.filter_freq) { const double fr = static_cast<double>(g.occ_cnt) / total; if (fr > P.freq_thr) continue; }     // seed_freq_filter
        if (W.n_mems >= AC_MAX_MEMS) { W.overflow = 1; return false; }
        ac_mem_t& M = W.mems[W.n_mems++];
        M.pos = g.pos; M.len = g.len; M.idx = g.idx; M.rpos = g.rpos; M.mate = g.mate; M.occs = occs + g.occ_off; M.nocc = g.occ_cnt;
    }
    W.prof[0] += AC_CLOCK() - pc0;
    return ac_chain(W, P);
}
// find_chains (chain.hpp:221-438) over W.mems[0 .. n_mems): anchors, chaining DP, chain starts, backtrack, chains by score.
// sec != nullptr: find_chains_secondary (chain.hpp:442-727, -Z): per anchor also the best predecessor that does not lie on the primary chain of the current best
// one; the secondary chains are added behind the primary ones before the sort by score, and the chain starts of both tracks are sorted by score alone.
AC_HD_BIG bool ac_chain(ac_ws_t& W, const ac_params_t& P, ac_sec_t* sec) {
    unsigned long long pc0 = AC_CLOCK();
    size_t tot_mem_length = 0, na = 0;
    for (uint32_t i = 0; i < W.n_mems; ++i) { na += W.mems[i].nocc; tot_mem_length += (size_t)W.mems[i].len * W.mems[i].nocc; }
    if (na == 0) return false;
    if (na > AC_MAX_ANCH) { W.overflow = 1; return false; }
    for (uint32_t i = 0; i < W.n_mems; ++i)
        for (uint32_t j = 0; j < W.mems[i].nocc; ++j) { ac_anchor_t& A = W.anch[W.n_anch++]; A.mem = i; A.occ = j; A.x = W.mems[i].occs[j] + W.mems[i].len - 1; }
    const float avg_mem_length = (float)tot_mem_length / na;
    { const unsigned long long x = AC_CLOCK(); W.prof[0] += x - pc0; pc0 = x; }
    lsort::sort(W.anch, (long)na, [](const ac_anchor_t& x, const ac_anchor_t& y) { return x.x < y.x; }, W.sort_stack);
    { const unsigned long long x = AC_CLOCK(); W.prof[1] += x - pc0; pc0 = x; }
    // one 32-byte node per anchor: everything the chaining loops read or write about it (position, read coordinates, mate
    // and the f / p / t / msc arrays of chain.hpp:254-263) comes with one request
    for (size_t i = 0; i < na; ++i) {
        const ac_mem_t& m = W.mems[W.anch[i].mem];
        ac_node_t nd;
        nd.x = W.anch[i].x; nd.rpos = (uint16_t)m.rpos; nd.len = (uint16_t)m.len; nd.mate = (uint8_t)m.mate; nd.pad0 = 0; nd.pad1 = 0;
        nd.f = 0; nd.p = 0; nd.t = 0; nd.msc = 0;               // std::vector<ll> t(n, 0)
        W.node[i] = nd;
        if (sec) { sec[i].f = 0; sec[i].p = 0; sec[i].t = 0; sec[i].msc = 0; }
    }
    long long lb = 0;
    for (size_t i = 0; i < na; ++i) {
        const ac_node_t ni = W.node[i];
        const long long x_i = (long long)ni.x, y_i = ni.rpos, w_i = ni.len;
        const uint32_t mate_i = ni.mate;
        long long max_f = w_i, max_j = -1, max_sec_f = w_i, max_sec_j = -1;
        size_t n_pred = 0;
        if (i - (size_t)lb > (size_t)P.max_iter) lb = (long long)i - P.max_iter;
        for (long long j = (long long)i - 1; j >= lb; --j) {
            const ac_node_t nj = W.node[j];
            const long long x_j = (long long)nj.x, y_j = nj.rpos;
            const uint32_t mate_j = nj.mate;
            if (mate_i != mate_j && ((mate_i ^ mate_j) != 3)) continue;
            if (x_i > x_j + P.max_dist_x) { lb = j; continue; }
            const long long x_d = x_i - x_j, y_d = y_i - y_j;
            const int32_t l = (int32_t)(y_d > x_d ? (y_d - x_d) : (x_d - y_d));
            const uint32_t ilog_l = l > 0 ? (uint32_t)ac_ilog2((uint32_t)l) : 0;
            if (mate_i == mate_j && (y_j >= y_i || y_d > P.max_dist_y)) continue;
            const long long mn = y_d < x_d ? y_d : x_d;
            const long long alpha = mn < w_i ? mn : w_i;
            long long beta = 0;
            if (mate_i != mate_j) {
                if (x_d == 0) ++beta;
                else { const int c_lin = (int)(l * .01 * avg_mem_length); beta = c_lin < (long long)ilog_l ? c_lin : (long long)ilog_l; }
            } else {
                beta = l > 0 ? ((long long)(.01 * l * avg_mem_length) + ilog_l) >> 1 : 0;
            }
            const long long score = nj.f + (alpha - beta);
            if (score > max_f) { max_f = score; max_j = j; if (n_pred > 0) --n_pred; }
            else if (sec && (long long)sec[j].f + (alpha - beta) > max_sec_f) {          // chain.hpp:586-612
                if (max_j >= 0) {
                    const uint64_t pos_j = nj.x - nj.len + 1;                            // the occurrence the anchor stands for
                    bool uniq = true;
                    for (long long tmp = max_j; tmp >= 0; tmp = W.node[tmp].p) if (W.node[tmp].x - W.node[tmp].len + 1 == pos_j) { uniq = false; break; }
                    if (uniq) { max_sec_f = (long long)sec[j].f + (alpha - beta); max_sec_j = j; }
                }
            }
            else if ((size_t)(long long)nj.t == i && (++n_pred > (size_t)P.max_pred)) break;
            if (nj.p > 0) W.node[nj.p].t = (int32_t)i;
            if (sec && sec[j].p > 0) sec[sec[j].p].t = (int32_t)i;
        }
        W.node[i].f = (int32_t)max_f; W.node[i].p = (int32_t)max_j;
        W.node[i].msc = (max_j >= 0 && W.node[max_j].msc > max_f) ? W.node[max_j].msc : (int32_t)max_f;
        if (sec) { sec[i].f = (int32_t)max_sec_f; sec[i].p = (int32_t)max_sec_j; sec[i].msc = (max_sec_j >= 0 && sec[max_sec_j].msc > max_sec_f) ? sec[max_sec_j].msc : (int32_t)max_sec_f; }
    }
    { const unsigned long long x = AC_CLOCK(); W.prof[2] += x - pc0; pc0 = x; }
    for (size_t i = 0; i < na; ++i) W.node[i].t = 0;
    for (size_t i = 0; i < na; ++i) if (W.node[i].p >= 0) W.node[W.node[i].p].t = 1;
    uint32_t ns = 0;
    for (size_t i = 0; i < na; ++i) {
        if (W.node[i].t == 0 && W.node[i].msc > P.min_chain_score) {
            size_t j = i;
            while (W.node[j].f < W.node[j].msc) j = (size_t)W.node[j].p;
            if (ns >= AC_MAX_CHAINS) { W.overflow = 1; return false; }
            W.starts[ns].f = W.node[j].f; W.starts[ns].j = j; ++ns;
        }
    }
    if (ns == 0) return false;
    if (sec) lsort::sort(W.starts, (long)ns, [](const ac_start_t& x, const ac_start_t& y) { return x.f > y.f; }, W.sort_stack);      // chain_start_cmp (chain.hpp:663-668): the score alone
    else lsort::sort(W.starts, (long)ns, [](const ac_start_t& x, const ac_start_t& y) { return x.f > y.f || (x.f == y.f && x.j > y.j); }, W.sort_stack);   // std::greater<pair>
    for (size_t i = 0; i < na; ++i) W.node[i].t = 0;
    for (uint32_t i = 0; i < ns; ++i) {
        long long j = (long long)W.starts[i].j;
        ac_chain_t c;
        c.mate = W.node[j].mate;
        c.score = W.starts[i].f;
        c.off = W.pool_used; c.cnt = 0; c.paired = 0;
        do {
            c.paired |= (c.mate != (uint32_t)W.node[j].mate) ? 1u : 0u;
            if (W.pool_used >= AC_MAX_POOL) { W.overflow = 1; return false; }
            W.pool[W.pool_used++] = (uint32_t)j; c.cnt++;
            W.node[j].t = 1; j = W.node[j].p;
        } while (j >= 0 && W.node[j].t == 0);
        bool keep = false;
        if (j < 0) keep = (long long)c.cnt >= P.min_chain_length;
        else if (W.starts[i].f - W.node[j].f >= P.min_chain_score) keep = (long long)c.cnt >= P.min_chain_length;
        if (keep) W.chains[W.n_chains++] = c;            // (a dropped chain leaves its anchors in the pool; harmless)
    }
    if (sec) {          // the secondary track: chain ends, starts (their own sort), backtrack over p_sec / f_sec; the chains go behind the primary ones
        for (size_t i = 0; i < na; ++i) sec[i].t = 0;
        for (size_t i = 0; i < na; ++i) if (sec[i].p >= 0) sec[sec[i].p].t = 1;
        uint32_t ns2 = 0;
        for (size_t i = 0; i < na; ++i) {
            if (sec[i].t == 0 && sec[i].msc > P.min_chain_score) {
                size_t j = i;
                while (sec[j].f < sec[j].msc) j = (size_t)sec[j].p;
                if (ns2 >= AC_MAX_CHAINS) { W.overflow = 1; return false; }
                W.starts[ns2].f = sec[j].f; W.starts[ns2].j = j; ++ns2;
            }
        }
        lsort::sort(W.starts, (long)ns2, [](const ac_start_t& x, const ac_start_t& y) { return x.f > y.f; }, W.sort_stack);
        for (size_t i = 0; i < na; ++i) sec[i].t = 0;
        for (uint32_t i = 0; i < ns2; ++i) {
            long long j = (long long)W.starts[i].j;
            ac_chain_t c;
            c.mate = W.node[j].mate;
            c.score = W.starts[i].f;
            c.off = W.pool_used; c.cnt = 0; c.paired = 0;
            do {
                c.paired |= (c.mate != (uint32_t)W.node[j].mate) ? 1u : 0u;
                if (W.pool_used >= AC_MAX_POOL) { W.overflow = 1; return false; }
                W.pool[W.pool_used++] = (uint32_t)j; c.cnt++;
                sec[j].t = 1; j = sec[j].p;
            } while (j >= 0 && sec[j].t == 0);
            bool keep = false;
            if (j < 0) keep = (long long)c.cnt >= P.min_chain_length;
            else if (W.starts[i].f - sec[j].f >= P.min_chain_score) keep = (long long)c.cnt >= P.min_chain_length;
            if (keep) { if (W.n_chains >= AC_MAX_CHAINS) { W.overflow = 1; return false; } W.chains[W.n_chains++] = c; }
        }
    }
    lsort::sort(W.chains, (long)W.n_chains, [](const ac_chain_t& x, const ac_chain_t& y) { return x.score > y.score; }, W.sort_stack);
    for (uint32_t i = 0; i < W.n_chains; ++i) W.score_cache[i] = INT32_MIN;
    { const unsigned long long x = AC_CLOCK(); W.prof[3] += x - pc0; pc0 = x; }
    W.stage = AC_LOOP;
    return true;
}

// ---- fill_chain, part 1: define the DP problems (aligner_ksw2.hpp:2782-2979) ----
AC_HD void ac_task(ac_ws_t& W, uint64_t q_off, int qlen, int qmode, uint64_t t_off, int tlen, int tmode, int flag, int32_t& id) {
    moni_dp_task_t& t = W.tasks[W.n_tasks];
    t.q_off = q_off; t.t_off = t_off; t.qlen = qlen; t.tlen = tlen; t.flag = flag; t.reserved = DP_Q_READS | DP_T_TEXT | qmode | tmode;
    id = (int32_t)W.n_tasks++;
}
// query segment R[a .. a+len) of the strand-oriented read, optionally reversed
AC_HD void ac_qseg(uint64_t off, uint64_t m, uint32_t strand, uint64_t a, uint64_t len, bool reversed, uint64_t& q_off, int& qmode) {
    if (!strand) { q_off = reversed ? off + a + len - 1 : off + a; qmode = reversed ? DP_Q_REV : 0; }
    else { q_off = reversed ? off + (m - (a + len)) : off + (m - 1 - a); qmode = DP_Q_COMP | (reversed ? 0 : DP_Q_REV); }
    if (len == 0) q_off = off;
}

// returns false if the chain does not fit (overflow)
// F.an_mem / F.an_occ [0 .. F.n_an): the anchors left to right; the read is reads[off .. off + m_); tasks are appended to W.tasks
AC_HD_BIG bool ac_fill_begin_g(ac_ws_t& W, const ac_params_t& P, ac_fill_t& F, uint64_t off, uint32_t m_, bool score_only) {
    F.score_only = score_only; F.overlap = 0;
    F.t_lc = F.t_rc = F.t_glob = -1; F.lc_mqe_t = F.rc_mqe_t = -1; F.score = 0; F.score_pos = 0;
    const ac_mem_t& first = W.mems[F.an_mem[0]];
    const ac_mem_t& last = W.mems[F.an_mem[F.n_an - 1]];
    F.strand = (first.mate & 2) ? 1 : 0;
    const uint64_t m = m_, ext_len = P.ext_len, n = P.n_text;
    F.lcs_len = first.idx;
    F.rcs_occ = (uint64_t)last.idx + last.len;
    F.rcs_len = m - F.rcs_occ;
    const int ext_flag = score_only ? DP_EZ_SCORE_ONLY : (DP_EZ_EXTZ_ONLY | DP_EZ_RIGHT);
    const uint64_t mem_pos = ac_occ(W, F.an_mem[0], F.an_occ[0]);
    if (F.lcs_len > 0) {
        const uint64_t lc_occ = mem_pos > ext_len ? mem_pos - ext_len : 0;
        const uint64_t lc_len = mem_pos > ext_len ? ext_len : ext_len - mem_pos;     // sic (aligner_ksw2.hpp:2796)
        uint64_t q_off; int qmode;
        ac_qseg(off, m, F.strand, 0, F.lcs_len, true, q_off, qmode);
        ac_task(W, q_off, (int)F.lcs_len, qmode, lc_len ? lc_occ + lc_len - 1 : 0, (int)lc_len, DP_T_REV, ext_flag, F.t_lc);
    }
    if (F.rcs_len > 0) {
        const uint64_t rc_occ = ac_occ(W, F.an_mem[F.n_an - 1], F.an_occ[F.n_an - 1]) + last.len;
        const uint64_t rc_len = rc_occ < n - ext_len ? ext_len : n - rc_occ;
        uint64_t q_off; int qmode;
        ac_qseg(off, m, F.strand, F.rcs_occ, F.rcs_len, false, q_off, qmode);
        ac_task(W, q_off, (int)F.rcs_len, qmode, rc_occ, (int)rc_len, 0, ext_flag, F.t_rc);
    }
    uint64_t last_ref = mem_pos + first.len, last_seq = (uint64_t)first.idx + first.len;
    for (uint32_t k = 1; k < F.n_an && !F.overlap; ++k) {       // aligner_ksw2.hpp:2888-2900
        const ac_mem_t& mk = W.mems[F.an_mem[k]];
        const uint64_t ref_occ = ac_occ(W, F.an_mem[k], F.an_occ[k]), seq_occ = mk.idx;
        if (last_ref > ref_occ || last_seq > seq_occ) F.overlap = 1;
        last_ref = ref_occ + mk.len; last_seq = seq_occ + mk.len;
    }
    for (uint32_t k = 0; k + 1 < F.n_an; ++k) { F.t_gap[k] = -1; F.gap_score[k] = 0; F.gap_cig[k] = 0; }
    if (!F.overlap) {
        last_ref = mem_pos + first.len; last_seq = (uint64_t)first.idx + first.len;
        for (uint32_t k = 1; k < F.n_an; ++k) {
            const ac_mem_t& mk = W.mems[F.an_mem[k]];
            const ac_mem_t& mp = W.mems[F.an_mem[k - 1]];
            const uint64_t ref_occ = ac_occ(W, F.an_mem[k], F.an_occ[k]), seq_occ = mk.idx;
            if (last_ref == ref_occ) {
                if (last_seq < seq_occ) {                                              // pure insertion
                    const uint64_t l = seq_occ - last_seq;
                    const uint64_t c1 = (uint64_t)(int64_t)P.gapo + l * (uint64_t)(int64_t)P.gape, c2 = (uint64_t)(int64_t)P.gapo2 + l * (uint64_t)(int64_t)P.gape2;
                    F.gap_score[k - 1] = (int32_t)(0 - (c1 < c2 ? c1 : c2));
                    F.gap_cig[k - 1] = (uint32_t)((l << 4) | 1);
                }
            } else if (last_seq == seq_occ) {                                          // "deletion": l is computed as 0 (aligner_ksw2.hpp:2939)
                const uint64_t l = seq_occ - last_seq;
                const uint64_t c1 = (uint64_t)(int64_t)P.gapo + l * (uint64_t)(int64_t)P.gape, c2 = (uint64_t)(int64_t)P.gapo2 + l * (uint64_t)(int64_t)P.gape2;
                F.gap_score[k - 1] = (int32_t)(0 - (c1 < c2 ? c1 : c2));
                F.gap_cig[k - 1] = (uint32_t)((l << 4) | 2);
                F.t_gap[k - 1] = -2;                                                   // has a one-op CIGAR even though its length is 0
            } else {
                const uint64_t cc_occ = ac_occ(W, F.an_mem[k - 1], F.an_occ[k - 1]) + mp.len;
                const uint64_t cc_len = ref_occ - cc_occ;
                const uint64_t ccs_pos = (uint64_t)mp.idx + mp.len;
                const uint64_t ccs_len = seq_occ - ccs_pos;
                uint64_t q_off; int qmode;
                ac_qseg(off, m, F.strand, ccs_pos, ccs_len, false, q_off, qmode);
                ac_task(W, q_off, (int)ccs_len, qmode, cc_occ, (int)cc_len, 0, DP_EZ_RIGHT, F.t_gap[k - 1]);
            }
            last_ref = ref_occ + mk.len; last_seq = seq_occ + mk.len;
        }
    }
    return true;
}
AC_HD_BIG bool ac_fill_begin(ac_ws_t& W, const ac_params_t& P, const ac_chain_t& ch, bool score_only) {
    ac_fill_t& F = W.fill;
    W.n_tasks = 0;
    if (ch.cnt > AC_MAX_FILL) { W.overflow = 1; return false; }
    F.n_an = ch.cnt;
    for (uint32_t k = 0; k < ch.cnt; ++k) {                  // stored right to left (chain.hpp:166-200); fill_chain wants left to right
        const ac_anchor_t& A = W.anch[W.pool[ch.off + ch.cnt - 1 - k]];
        F.an_mem[k] = A.mem; F.an_occ[k] = A.occ;
    }
    return ac_fill_begin_g(W, P, F, W.off, W.m, score_only);
}

// ---- fill_chain, part 2 (aligner_ksw2.hpp:2852-2886, 2975-2996); returns true if a dependent global problem was queued (appended to W.tasks) ----
AC_HD_BIG bool ac_fill_after_ext_g(ac_ws_t& W, const ac_params_t& P, ac_fill_t& F, uint64_t off, uint32_t m_, const moni_dp_result_t* res) {
    const ac_mem_t& last = W.mems[F.an_mem[F.n_an - 1]];
    int score_lc = 0, score_rc = 0;
    if (F.t_lc >= 0) { score_lc = res[F.t_lc].mqe; F.lc_mqe_t = res[F.t_lc].mqe_t; }
    if (F.t_rc >= 0) { score_rc = res[F.t_rc].mqe; F.rc_mqe_t = res[F.t_rc].mqe_t; }
    F.score = (int32_t)((uint32_t)score_lc + (uint32_t)score_rc);
    const uint64_t mem_pos = ac_occ(W, F.an_mem[0], F.an_occ[0]);
    const uint64_t mem_len = ac_occ(W, F.an_mem[F.n_an - 1], F.an_occ[F.n_an - 1]) + last.len - mem_pos;
    const uint64_t lq = (uint64_t)(int64_t)(F.lcs_len > 0 ? F.lc_mqe_t + 1 : 0);
    const uint64_t rq = (uint64_t)(int64_t)(F.rcs_len > 0 ? F.rc_mqe_t + 1 : 0);
    F.ref_pos = lq > mem_pos ? 0 : mem_pos - lq;
    F.ref_len = lq + mem_len + rq;
    F.score_pos = F.ref_pos;
    if (!F.overlap) {
        uint32_t sc = (uint32_t)F.score;
        for (uint32_t k = 1; k < F.n_an; ++k) {
            const int32_t gs = F.t_gap[k - 1] >= 0 ? res[F.t_gap[k - 1]].score : F.gap_score[k - 1];
            sc += (uint32_t)((uint64_t)W.mems[F.an_mem[k - 1]].len * (uint64_t)(int64_t)P.smatch + (uint64_t)(int64_t)gs);
        }
        sc += (uint32_t)((uint64_t)last.len * (uint64_t)(int64_t)P.smatch);
        F.score = (int32_t)sc;
        return false;
    }
    // overlapping MEMs: one global alignment of the whole read against the window (aligner_ksw2.hpp:2984-2996, 3009-3015)
    uint64_t q_off; int qmode;
    if (!F.strand) { q_off = off; qmode = 0; } else { q_off = off + m_ - 1; qmode = DP_Q_REV | DP_Q_COMP; }
    ac_task(W, q_off, (int)m_, qmode, F.ref_pos, (int)F.ref_len, 0, F.score_only ? DP_EZ_SCORE_ONLY : DP_EZ_RIGHT, F.t_glob);
    return true;
}
AC_HD_BIG bool ac_fill_after_ext(ac_ws_t& W, const ac_params_t& P, const moni_dp_result_t* res) {
    W.n_tasks = 0;                                 // (after the results were read: ac_fill_after_ext_g reads res, not W.tasks)
    return ac_fill_after_ext_g(W, P, W.fill, W.off, W.m, res);
}

// ---- fill_chain, part 3 (final pass): the stitched CIGAR (aligner_ksw2.hpp:3000-3108) ----
AC_HD_BIG bool ac_fill_final_g(ac_ws_t& W, const ac_params_t& P, ac_fill_t& F, const moni_dp_result_t* res, const uint32_t* cig, uint32_t* out, uint32_t& n_out) {
    n_out = 0;
    if (!ac_valid(P, F.ref_pos, F.ref_len)) return true;
    auto push = [&](uint32_t op) -> bool { if (n_out >= AC_MAX_CIGAR) { W.overflow = 1; return false; } out[n_out++] = op; return true; };
    auto push_merge_first = [&](const uint32_t* c, uint32_t n) -> bool {
        if (n > 0) { if ((c[0] & 0xf) == 0 && n_out > 0) out[n_out - 1] += c[0]; else if (!push(c[0])) return false; }
        for (uint32_t k = 1; k < n; ++k) if (!push(c[k])) return false;
        return true;
    };
    if (F.overlap) {
        const moni_dp_result_t& r = res[F.t_glob];
        for (uint32_t k = 0; k < r.n_cigar; ++k) if (!push(cig[r.cigar_off + k])) return false;
        F.score = r.score;
        return true;
    }
    if (F.t_lc >= 0) { const moni_dp_result_t& r = res[F.t_lc]; for (uint32_t k = 0; k < r.n_cigar; ++k) if (!push(cig[r.cigar_off + r.n_cigar - 1 - k])) return false; }
    for (uint32_t j = 0; j < F.n_an; ++j) {
        const uint32_t mlen = W.mems[F.an_mem[j]].len;
        if (n_out > 0 && (out[n_out - 1] & 0xf) == 0) out[n_out - 1] += mlen << 4;
        else if (!push(mlen << 4)) return false;
        if (j + 1 < F.n_an) {
            if (F.t_gap[j] >= 0) { const moni_dp_result_t& r = res[F.t_gap[j]]; if (!push_merge_first(cig + r.cigar_off, r.n_cigar)) return false; }
            else if (F.gap_cig[j] != 0 || F.t_gap[j] == -2) { const uint32_t op = F.gap_cig[j]; if (!push_merge_first(&op, 1)) return false; }
        }
    }
    if (F.t_rc >= 0) { const moni_dp_result_t& r = res[F.t_rc]; if (!push_merge_first(cig + r.cigar_off, r.n_cigar)) return false; }
    return true;
}
AC_HD_BIG bool ac_fill_final(ac_ws_t& W, const ac_params_t& P, const moni_dp_result_t* res, const uint32_t* cig) {
    return ac_fill_final_g(W, P, W.fill, res, cig, W.cigar, W.n_cigar);
}

// aligner_ksw2.hpp:553-597
AC_HD_BIG bool ac_check_left_mem(ac_ws_t& W, const ac_params_t& P, uint64_t ci) {
    const ac_chain_t& ch = W.chains[ci];
    const ac_anchor_t& A = W.anch[W.pool[ch.off + ch.cnt - 1]];                    // leftmost anchor
    const uint64_t left_ref = ac_seq_off(P, ac_lift(P, ac_occ(W, A.mem, A.occ))) + 1;  // index(lift(pos)).second + 1
    bool seen = false;
    for (uint32_t k = 0; k < W.n_left; ++k) {
        const uint64_t d = W.left[k].ref > left_ref ? W.left[k].ref - left_ref : left_ref - W.left[k].ref;
        if (d < P.region_dist && W.left[k].score == (uint64_t)ch.score) seen = true;
    }
    if (seen) return true;
    if (W.n_left >= AC_MAX_LEFT) { W.overflow = 1; return false; }
    W.left[W.n_left].ref = left_ref; W.left[W.n_left].score = (uint64_t)ch.score; ++W.n_left;
    return false;
}

// a scored chain comes back into the selection loop (aligner_ksw2.hpp:436-460, 528-548)
AC_HD_BIG void ac_absorb(ac_ws_t& W, const ac_params_t& P, int32_t score, uint64_t pos) {
    const uint64_t lft = ac_lift(P, pos);                                            // idx.lift(score.pos)
    if (score > W.max_score) { W.max_score = score; W.n_alt = 0; }
    else if (score == W.max_score) {
        if (W.n_alt >= AC_MAX_ALT) { W.overflow = 1; return; }
        W.alt_pos[W.n_alt] = pos; W.alt_score[W.n_alt] = score; ++W.n_alt;
    }
    bool replaced = false;
    for (uint32_t j = 0; j < W.n_best; ++j) {
        const uint64_t bl = W.best[j].lft;
        const uint64_t d = bl > lft ? bl - lft : lft - bl;
        if (d < P.region_dist) {
            if (score > W.best[j].score) {
                if (replaced) { W.best[j].score = 0; W.best[j].lft = 0; W.best[j].idx = W.i - 1; }
                else { W.best[j].score = score; W.best[j].lft = lft; W.best[j].idx = W.i; W.i++; replaced = true; }
            } else { j = W.n_best; replaced = true; W.i++; }
        }
    }
    if (!replaced) {
        if (W.n_best >= AC_MAX_BEST) { W.overflow = 1; return; }
        W.best[W.n_best].score = score; W.best[W.n_best].lft = lft; W.best[W.n_best].idx = W.i; ++W.n_best; W.i++;
    }
}

// Runs the selection loop until the read needs DP results (W.n_tasks > 0 or a fill without DP) or is done.
// Returns true if a fill was started (stage AC_WAIT_A / AC_FINAL_WAIT_A).
AC_HD_BIG bool ac_advance(ac_ws_t& W, const ac_params_t& P) {
    while (W.stage == AC_LOOP && !W.overflow) {
        if (W.i < W.n_chains && W.n_diff < P.check_k) {
            { const uint64_t v = (uint64_t)W.chains[W.i].score; bool f = false; for (uint32_t q = 0; q < W.n_diff; ++q) f = f || W.diff[q] == v; if (!f) W.diff[W.n_diff++] = v; }
            if (P.left_mem_check && ac_check_left_mem(W, P, W.i)) { ++W.i; continue; }
            if (W.overflow) return false;
            if (W.n_diff < P.check_k) {
                if (!ac_fill_begin(W, P, W.chains[W.i], true)) return false;
                W.stage = AC_WAIT_A;
                return true;
            }
            continue;
        }
        // after the loop (aligner_ksw2.hpp:464-509)
        while (W.n_best < 2) { W.best[W.n_best].score = 0; W.best[W.n_best].lft = 0; W.best[W.n_best].idx = W.n_chains; ++W.n_best; }
        lsort::sort(W.best, (long)W.n_best, [](const ac_best_t& x, const ac_best_t& y) {
            return x.score > y.score || (x.score == y.score && (x.lft > y.lft || (x.lft == y.lft && x.idx > y.idx)));        // std::greater<tuple>
        }, W.sort_stack);
        if (W.best[0].score < W.min_score) { W.stage = AC_DONE; return false; }
        W.score2 = W.best[1].score;
        W.final_chain = W.best[0].idx;
        if (W.final_chain >= W.n_chains || W.score_cache[W.final_chain] < W.min_score) { W.stage = AC_DONE; return false; }
        if (!ac_fill_begin(W, P, W.chains[W.final_chain], false)) return false;
        W.stage = AC_FINAL_WAIT_A;
        return true;
    }
    return false;
}

// Drive the read: consume the results of the DP problems it queued last (res / cig index the tasks of W.tasks in order) and
// continue until it queues new ones (returns with W.n_tasks > 0) or finishes (W.stage == AC_DONE) or overflows.
AC_HD_BIG void ac_drive(ac_ws_t& W, const ac_params_t& P, const moni_dp_result_t* res, const uint32_t* cig) {
    while (!W.overflow) {
        switch (W.stage) {
            case AC_LOOP:
                W.n_tasks = 0;
                if (!ac_advance(W, P)) return;              // done / overflow
                if (W.n_tasks > 0) return;                  // wait for DP
                break;                                      // a fill that needs no DP: fall through as if results had arrived
            case AC_WAIT_A:
            case AC_WAIT_B:
                if (W.stage == AC_WAIT_A) {
                    if (ac_fill_after_ext(W, P, res)) { W.stage = AC_WAIT_B; return; }
                } else W.fill.score = res[W.fill.t_glob].score;
                if (!ac_valid(P, W.fill.ref_pos, W.fill.ref_len)) W.fill.score = INT32_MIN;
                W.score_cache[W.i] = W.fill.score;
                ac_absorb(W, P, W.fill.score, W.fill.score_pos);
                W.stage = AC_LOOP;
                break;
            case AC_FINAL_WAIT_A:
                if (ac_fill_after_ext(W, P, res)) { W.stage = AC_FINAL_WAIT_B; return; }
                if (!ac_fill_final(W, P, res, cig)) return;
                W.aligned = 1; W.stage = AC_DONE; W.n_tasks = 0;
                return;
            case AC_FINAL_WAIT_B:
                if (!ac_fill_final(W, P, res, cig)) return;
                W.aligned = 1; W.stage = AC_DONE; W.n_tasks = 0;
                return;
            default:
                W.n_tasks = 0;
                return;
        }
    }
}
