// Host side of the full single-end path: everything of aligner::align that is not a kernel.
//
//   seeds (GPU)  ->  frequency filter, chaining            (aligner_ksw2.hpp:1905-1933, chain.hpp:221-438)
//                ->  chain-selection loop                  (aligner_ksw2.hpp:394-474, 528-597)
//                      each fill_chain = a few DP problems (aligner_ksw2.hpp:2752-3000)  -> extz_kernel batches
//                ->  final fill_chain: CIGAR stitching, MD/NM, lift, MAPQ, SAM text
//                                                          (aligner_ksw2.hpp:3000-3196, 2018-2098; mapq.hpp:146-184;
//                                                           sam.hpp:144-188,249-287)
//
// The reference runs this per read, sequentially, calling ksw2 inline.  Here every read is a small state machine that
// stops whenever it needs DP results; all reads of a batch advance together, and each round's DP problems (left/right
// extensions and gap fills of one fill_chain per read, or the dependent global realignment) go to the GPU as one
// batch whose operands are named by position in the resident read batch and the index text (no sequence copies).
// The decision logic has to be the reference's, quirk for quirk, because SAM must be identical; the structure
// (resumable per-read state, batch rounds, cached score-only results, operands by reference) is not.
//
// The DP/seed provider is abstract so that tests/host_sim can drive the same code with CPU stand-ins; the product
// wires it to the HIP kernels only (moni_align_batch in moni_hip.hip).
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/moni_hip.h"
#include "lift_core.h"

#ifndef DP_EZ_SCORE_ONLY
#define DP_EZ_SCORE_ONLY 0x01
#define DP_EZ_RIGHT 0x02
#define DP_EZ_EXTZ_ONLY 0x40
#define DP_Q_READS 0x01
#define DP_Q_REV 0x02
#define DP_Q_COMP 0x04
#define DP_T_TEXT 0x08
#define DP_T_REV 0x10
#define DP_NEG_INF (-0x40000000)
#endif

namespace mh {

typedef long long ll;

struct HostIndex {                 // host copies of what the host stages read (seqidx + text)
    uint64_t n_text = 0, w = 0;
    std::vector<uint64_t> seq_starts;
    std::vector<std::string> names;
    const uint8_t* text = nullptr;
    std::vector<moni_lift_seq_t> lift_seqs;        // liftidx::lifts in the run form of lift_core.h (lift_build.hpp)
    std::vector<moni_lift_run_t> lift_runs;

    size_t rank1(uint64_t i) const { return (size_t)(std::lower_bound(seq_starts.begin(), seq_starts.end(), i) - seq_starts.begin()); }
    uint64_t select1(size_t k) const { return seq_starts[k - 1]; }
    std::pair<size_t, uint64_t> index(uint64_t pos) const {            // seqidx.hpp:149-154
        size_t rk = rank1(pos + 1);
        return std::make_pair(rk - 1, pos - select1(rk));
    }
    bool valid(uint64_t pos, uint64_t len) const { return pos + len <= select1(rank1(pos + 1) + 1); }   // seqidx.hpp:164-167
    uint64_t lift(uint64_t pos) const {                                // liftidx.hpp:89-95
        const size_t rk = rank1(pos + 1);
        const moni_lift_seq_t& L = lift_seqs[rk - 1];
        return L.second + lift_pos(lift_runs.data() + L.run_off, L.n_runs, pos - select1(rk));
    }
    // liftidx.hpp:159-164: the CIGAR of an alignment that starts at text position pos, lifted
    void lift_cigar_at(uint64_t pos, const uint32_t* cig, uint32_t n_cig, std::vector<uint32_t>& out) const {
        const size_t rk = rank1(pos + 1);
        const moni_lift_seq_t& L = lift_seqs[rk - 1];
        uint64_t units = 0;
        for (uint32_t k = 0; k < n_cig; ++k) units += cig[k] >> 4;
        // an alignment crosses few runs: try a small buffer first (the full bound is two entries per run of the sequence: tens of KB to clear)
        out.resize(2 * (size_t)n_cig + 64);
        int n = lift_cigar(lift_runs.data() + L.run_off, L.n_runs, pos - select1(rk), cig, n_cig, out.data(), (uint32_t)out.size());
        if (n < 0) {
            out.resize(2 * (size_t)n_cig + 2 * (size_t)L.n_runs + 4);
            n = lift_cigar(lift_runs.data() + L.run_off, L.n_runs, pos - select1(rk), cig, n_cig, out.data(), (uint32_t)out.size());
        }
        out.resize(n < 0 ? 0 : (size_t)n);
        (void)units;
    }
    uint64_t seq_length(size_t i) const { return select1(i + 2) - select1(i + 1) - w; }
    std::string sam_header() const {                                   // aligner_ksw2.hpp:3213-3219, seqidx.hpp:174-180
        std::string res = "@HD\tVN:1.6\tSO:unknown\n";
        for (size_t i = 0; i < names.size(); ++i) res += "@SQ\tSN:" + names[i] + "\tLN:" + std::to_string(seq_length(i)) + "\n";
        res += "@PG\tID:moni\tPN:moni\tVN:0.1.0\n";
        return res;
    }
};

struct Backend {
    virtual ~Backend() {}
    // seeds of the resident batch, in the reference's per-read order
    virtual int seed(const moni_seed_params_t& p, std::vector<moni_mem_t>& mems, std::vector<uint64_t>& occs, std::vector<uint64_t>& read_mem_off) = 0;
    // DP problems; results[i].cigar_off indexes cig
    virtual int dp(const moni_dp_params_t& p, const std::vector<moni_dp_task_t>& tasks, std::vector<moni_dp_result_t>& res, std::vector<uint32_t>& cig) = 0;
};

static inline int ilog2_u32(uint32_t v) {                              // common.hpp:540-545 (floor(log2 v), v > 0)
    int r = 0;
    while (v >>= 1) ++r;
    return r;
}

static inline uint8_t nt4_of(uint8_t b) {                              // aligner_ksw2.hpp:3272-3288
    if (b < 4) return b;
    switch (b & 0xDF) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}
static inline uint8_t compl_of(uint8_t b) {                            // kpbseq.h:120-137
    switch (b) { case 'A': case 'a': return 'T'; case 'C': case 'c': return 'G'; case 'G': case 'g': return 'C'; case 'T': case 't': return 'A'; default: return b; }
}

struct Mem {                       // mem_t view (mems.hpp:31-60)
    uint64_t pos;
    uint32_t len, idx, rpos, mate;
    const uint64_t* occs;
    uint32_t nocc;
};
struct Chain {                     // chain_t (chain.hpp:25-52)
    ll score = 0;
    uint32_t mate = 2;
    std::vector<uint32_t> anchors;
};

// ---- chaining: chain.hpp:221-438 ---------------------------------------------------------------------------------------
static bool chain_mems(const std::vector<Mem>& mems, std::vector<std::pair<uint32_t, uint32_t>>& anchors, std::vector<Chain>& chains,
                       const moni_align_params_t& P) {
    size_t tot_mem_length = 0;
    for (size_t i = 0; i < mems.size(); ++i) {
        for (uint32_t j = 0; j < mems[i].nocc; ++j) anchors.push_back(std::make_pair((uint32_t)i, j));
        tot_mem_length += (size_t)mems[i].len * mems[i].nocc;
    }
    const float avg_mem_length = (float)tot_mem_length / anchors.size();
    auto xend = [&](const std::pair<uint32_t, uint32_t>& a) -> uint64_t { return mems[a.first].occs[a.second] + mems[a.first].len - 1; };
    std::sort(anchors.begin(), anchors.end(), [&](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return xend(a) < xend(b); });
    const size_t na = anchors.size();
    static thread_local std::vector<ll> f, p, msc, t;        // per-thread scratch, reused across reads
    f.assign(na, 0); p.assign(na, 0); msc.assign(na, 0); t.assign(na, 0);
    ll lb = 0;
    for (size_t i = 0; i < na; ++i) {
        const Mem& mi = mems[anchors[i].first];
        const ll x_i = (ll)xend(anchors[i]), y_i = mi.rpos, w_i = mi.len;
        const uint32_t mate_i = mi.mate;
        ll max_f = w_i, max_j = -1;
        size_t n_pred = 0;
        if (i - (size_t)lb > (size_t)P.max_iter) lb = (ll)i - P.max_iter;
        for (ll j = (ll)i - 1; j >= lb; --j) {
            const Mem& mj = mems[anchors[j].first];
            const ll x_j = (ll)xend(anchors[j]), y_j = mj.rpos;
            const uint32_t mate_j = mj.mate;
            if (mate_i != mate_j && ((mate_i ^ mate_j) != 3)) continue;
            if (x_i > x_j + P.max_dist_x) { lb = j; continue; }
            const ll x_d = x_i - x_j, y_d = y_i - y_j;
            const int32_t l = (int32_t)(y_d > x_d ? (y_d - x_d) : (x_d - y_d));
            const uint32_t ilog_l = l > 0 ? (uint32_t)ilog2_u32((uint32_t)l) : 0;
            if (mate_i == mate_j && (y_j >= y_i || y_d > P.max_dist_y)) continue;
            const ll alpha = std::min(std::min(y_d, x_d), w_i);
            ll beta = 0;
            if (mate_i != mate_j) {
                if (x_d == 0) ++beta;
                else { const int c_lin = (int)(l * .01 * avg_mem_length); beta = c_lin < (ll)ilog_l ? c_lin : (ll)ilog_l; }
            } else {
                beta = l > 0 ? ((ll)(.01 * l * avg_mem_length) + ilog_l) >> 1 : 0;
            }
            const ll score = f[j] + (alpha - beta);
            if (score > max_f) { max_f = score; max_j = j; if (n_pred > 0) --n_pred; }
            else if ((size_t)t[j] == i && (++n_pred > (size_t)P.max_pred)) break;
            if (p[j] > 0) t[p[j]] = (ll)i;
        }
        f[i] = max_f; p[i] = max_j;
        msc[i] = (max_j >= 0 && msc[max_j] > max_f) ? msc[max_j] : max_f;
    }
    std::fill(t.begin(), t.end(), 0);
    for (size_t i = 0; i < na; ++i) if (p[i] >= 0) t[p[i]] = 1;
    static thread_local std::vector<std::pair<ll, size_t>> starts;
    starts.clear();
    for (size_t i = 0; i < na; ++i) {
        if (t[i] == 0 && msc[i] > P.min_chain_score) {
            size_t j = i;
            while (f[j] < msc[j]) j = (size_t)p[j];
            starts.push_back(std::make_pair(f[j], j));
        }
    }
    if (starts.empty()) return false;
    std::sort(starts.begin(), starts.end(), std::greater<std::pair<ll, size_t>>());
    std::fill(t.begin(), t.end(), 0);
    for (size_t i = 0; i < starts.size(); ++i) {
        ll j = (ll)starts[i].second;
        Chain c;
        c.mate = mems[anchors[j].first].mate;
        c.score = starts[i].first;
        do { c.anchors.push_back((uint32_t)j); t[j] = 1; j = p[j]; } while (j >= 0 && t[j] == 0);
        if (j < 0) { if ((ll)c.anchors.size() >= P.min_chain_length) chains.push_back(std::move(c)); }
        else if (starts[i].first - f[j] >= P.min_chain_score) { if ((ll)c.anchors.size() >= P.min_chain_length) chains.push_back(std::move(c)); }
    }
    std::sort(chains.begin(), chains.end(), [](const Chain& a, const Chain& b) { return a.score > b.score; });
    return true;
}

// ---- MAPQ: mapq.hpp:146-184 ------------------------------------------------------------------------------------------------
static size_t mapq_se_bwa(int32_t score, int32_t score2, int32_t rlen, int32_t qlen, int32_t min_seed_length, int32_t match_score,
                          int32_t mismatch_score, double coeff_len, int32_t coeff_fac, int32_t sub_n = 0) {
    int32_t mapq = 0;
    const int32_t l = std::max(rlen, qlen);
    const int32_t sub = score2 ? score2 : min_seed_length * match_score;
    if (sub >= score) return mapq;
    const double identity = 1. - (double)(l * match_score - score) / (match_score + mismatch_score) / l;
    if (score == 0) mapq = 0;
    else {
        double tmp = l < coeff_len ? 1. : coeff_fac / log(l);
        tmp *= identity * identity;
        mapq = (int)(6.02 * (score - sub) / match_score * tmp * tmp + .499);
    }
    if (sub_n > 0) mapq -= (int)(4.343 * log(sub_n + 1) + .499);
    if (mapq > 60) mapq = 60;
    if (mapq < 0) mapq = 0;
    mapq = (int)(mapq * 1. + .499);
    return mapq;
}

// sam.hpp:249-287
static size_t md_core(const uint8_t* tseq, const uint8_t* qseq, const std::vector<uint32_t>& cigar, std::string& mdz) {
    int q_off = 0, t_off = 0, l_MD = 0, NM = 0;
    for (size_t i = 0; i < cigar.size(); ++i) {
        const int op = cigar[i] & 0xf, len = (int)(cigar[i] >> 4);
        if (op == 0 || op == 7 || op == 8) {
            for (int j = 0; j < len; ++j) {
                if (qseq[q_off + j] != tseq[t_off + j]) { mdz += std::to_string(l_MD); mdz.push_back("ACGTN"[tseq[t_off + j]]); l_MD = 0; ++NM; }
                else ++l_MD;
            }
            q_off += len; t_off += len;
        } else if (op == 1) { q_off += len; NM += len; }
        else if (op == 2) {
            mdz += std::to_string(l_MD); mdz.push_back('^');
            for (int j = 0; j < len; ++j) mdz.push_back("ACGTN"[tseq[t_off + j]]);
            l_MD = 0; t_off += len; NM += len;
        } else if (op == 3) t_off += len;
    }
    if (l_MD > 0) mdz += std::to_string(l_MD);
    return (size_t)NM;
}

// Small vector with inline storage: chains rarely have more than a few anchors, and a heap allocation per fill_chain
// member per read per round is what the host stages would otherwise spend their time on.
template <class Tp, int N>
struct SmallVec {
    Tp inl[N];
    std::vector<Tp> big;
    uint32_t n = 0;
    void assign(size_t k, const Tp& v) { n = (uint32_t)k; if (k > (size_t)N) big.assign(k, v); else for (size_t i = 0; i < k; ++i) inl[i] = v; }
    void resize(size_t k) { n = (uint32_t)k; if (k > (size_t)N) big.resize(k); }
    size_t size() const { return n; }
    Tp& operator[](size_t i) { return n > (uint32_t)N ? big[i] : inl[i]; }
    const Tp& operator[](size_t i) const { return n > (uint32_t)N ? big[i] : inl[i]; }
    Tp& back() { return (*this)[n - 1]; }
    const Tp& back() const { return (*this)[n - 1]; }
};

struct moni_alt_like { uint64_t pos; int32_t score; int32_t pad; };      // layout of moni_alt_t (align_kernel.hip)

struct Sam {                       // sam_t (sam.hpp:47-112), the fields the SE path sets
    bool rev_read = false;         // sam.read = &read_rev
    size_t flag = 4, pos = 0, mapq = 255;
    std::string rname = "*", cigar = "*";
    size_t as = 0, nm = 0, zs = 0;
    std::string md;
    std::vector<std::string> alt_haplotypes;
    std::vector<size_t> alt_pos, alt_scores;
    size_t rlen = 0;
    std::string lift_rname = "*", lift_cigar = "*";
    size_t lift_pos = 0, lift_nm = 0;
    std::string lift_md;
    bool unmapped_lft = false;
    std::string rnext = "*";       // paired-end only (pe_host.hpp)
    size_t pnext = 0;
    long long tlen = 0;
};

struct Score { int32_t score = 0; uint64_t pos = 0, lft = 0; bool unmapped_lft = false; };

// One fill_chain in flight (aligner_ksw2.hpp:2752-3196), split where it needs DP results.
struct Fill {
    bool score_only = true;
    SmallVec<std::pair<uint32_t, uint32_t>, 8> an;   // chain anchors, left to right
    uint32_t strand = 0;
    uint64_t lcs_len = 0, rcs_len = 0, rcs_occ = 0;
    bool overlap = false;
    int t_lc = -1, t_rc = -1, t_glob = -1;
    SmallVec<int, 8> t_gap;                          // DP task per gap, -1 for the closed-form shortcuts
    SmallVec<int32_t, 8> gap_score;                  // shortcut scores
    SmallVec<uint32_t, 8> gap_cig;                   // shortcut CIGAR op (0 = none)
    uint64_t ref_pos = 0, ref_len = 0;
    int32_t lc_mqe_t = -1, rc_mqe_t = -1;
    Score score;
};

// the MEM statistics of one read (`-c`: csv_t, include/common/csv.hpp:26-52)
struct CsvRead { size_t num_uniq_mems = 0, total_mem_occ = 0, high_occ_mem = 0, low_occ_mem = 0, num_mems_filter = 0, num_chains_skipped = 0; double max_mem_freq = 0, min_mem_freq = 1; };

struct ReadState {
    CsvRead csv;
    uint64_t off = 0; uint32_t m = 0;                // read bytes in the batch
    std::vector<Mem> mems;
    std::vector<std::pair<uint32_t, uint32_t>> anchors;
    std::vector<Chain> chains;
    int32_t min_score = 0;
    // selection loop (aligner_ksw2.hpp:394-474)
    size_t i = 0;
    SmallVec<size_t, 8> different_scores;           // std::set<size_t> of at most check_k chain scores
    std::vector<std::tuple<int32_t, size_t, size_t>> best_scores;
    std::vector<std::pair<size_t, size_t>> left_mem_vec;
    int32_t max_score = 0;
    std::vector<int32_t> chain_score_cache;          // score-only result per chain (INT32_MIN+1 = not computed)
    std::vector<uint64_t> chain_pos_cache;
    int32_t score2 = 0;
    Fill fill;
    enum Stage { LOOP, WAIT_A, WAIT_B, FINAL_WAIT_A, FINAL_WAIT_B, DONE } stage = LOOP;
    bool aligned = false;
    size_t final_chain = 0;
    Sam sam;
    // DP task bookkeeping for the current round
    uint32_t task_base = 0;
    int owner_thread = -1;
};

struct Aligner {
    const HostIndex& ix;
    moni_align_params_t P;
    const uint8_t* reads;            // host copy of the resident batch
    const uint64_t* offs;
    int32_t mapq_coeff_fac;
    size_t max_name_len = 1;

    Aligner(const HostIndex& ix_, const moni_align_params_t& p, const uint8_t* r, const uint64_t* o) : ix(ix_), P(p), reads(r), offs(o) {
        mapq_coeff_fac = (int32_t)log(50.0f);        // aligner_ksw2.hpp:3250-3251
        for (const auto& nm : ix.names) max_name_len = std::max(max_name_len, nm.size());
    }

    uint64_t occ_of(const ReadState& R, const std::pair<uint32_t, uint32_t>& a) const { return R.mems[a.first].occs[a.second]; }

    // ---- fill_chain, part 1: define the DP problems (aligner_ksw2.hpp:2782-2979) ----
    void fill_begin(ReadState& R, const Chain& chain, bool score_only, std::vector<moni_dp_task_t>& tasks) {
        Fill& F = R.fill;
        F = Fill();
        const size_t q0 = tasks.size();                  // task ids are relative to the read's first task of this round
        F.score_only = score_only;
        const size_t cn = chain.anchors.size();              // stored right to left (chain.hpp:166-200); fill_chain wants left to right
        F.an.resize(cn);
        for (size_t k = 0; k < cn; ++k) F.an[k] = R.anchors[chain.anchors[cn - 1 - k]];
        const Mem& first = R.mems[F.an[0].first];
        const Mem& last = R.mems[F.an.back().first];
        F.strand = (first.mate & 2) ? 1 : 0;
        const uint64_t m = R.m, ext_len = P.ext_len, n = ix.n_text;
        F.lcs_len = first.idx;
        F.rcs_occ = (uint64_t)last.idx + last.len;
        F.rcs_len = m - F.rcs_occ;
        const int ext_flag = score_only ? DP_EZ_SCORE_ONLY : (DP_EZ_EXTZ_ONLY | DP_EZ_RIGHT);
        auto add = [&](uint64_t q_off, int qlen, int qmode, uint64_t t_off, int tlen, int tmode, int flag) -> int {
            moni_dp_task_t t;
            t.q_off = q_off; t.t_off = t_off; t.qlen = qlen; t.tlen = tlen; t.flag = flag; t.reserved = DP_Q_READS | DP_T_TEXT | qmode | tmode;
            tasks.push_back(t);
            return (int)(tasks.size() - 1 - q0);
        };
        // query segment R[a .. a+len) of the strand-oriented read, optionally reversed
        auto qseg = [&](uint64_t a, uint64_t len, bool reversed, uint64_t& q_off, int& qmode) {
            if (!F.strand) { q_off = reversed ? R.off + a + len - 1 : R.off + a; qmode = reversed ? DP_Q_REV : 0; }
            else {           // R[x] = compl(read[m-1-x])
                q_off = reversed ? R.off + (m - (a + len)) : R.off + (m - 1 - a);
                qmode = DP_Q_COMP | (reversed ? 0 : DP_Q_REV);
            }
            if (len == 0) { q_off = R.off; }
        };
        if (F.lcs_len > 0) {
            const uint64_t mem_pos = occ_of(R, F.an[0]);
            const uint64_t lc_occ = mem_pos > ext_len ? mem_pos - ext_len : 0;
            const uint64_t lc_len = mem_pos > ext_len ? ext_len : ext_len - mem_pos;     // sic (aligner_ksw2.hpp:2796)
            uint64_t q_off; int qmode;
            qseg(0, F.lcs_len, true, q_off, qmode);
            F.t_lc = add(q_off, (int)F.lcs_len, qmode, lc_len ? lc_occ + lc_len - 1 : 0, (int)lc_len, DP_T_REV, ext_flag);
        }
        if (F.rcs_len > 0) {
            const uint64_t rc_occ = occ_of(R, F.an.back()) + last.len;
            const uint64_t rc_len = rc_occ < n - ext_len ? ext_len : n - rc_occ;
            uint64_t q_off; int qmode;
            qseg(F.rcs_occ, F.rcs_len, false, q_off, qmode);
            F.t_rc = add(q_off, (int)F.rcs_len, qmode, rc_occ, (int)rc_len, 0, ext_flag);
        }
        // overlap test (aligner_ksw2.hpp:2888-2900)
        const uint64_t mem_pos = occ_of(R, F.an[0]);
        uint64_t last_ref = mem_pos + first.len, last_seq = (uint64_t)first.idx + first.len;
        for (size_t k = 1; k < F.an.size() && !F.overlap; ++k) {
            const Mem& mk = R.mems[F.an[k].first];
            const uint64_t ref_occ = occ_of(R, F.an[k]), seq_occ = mk.idx;
            if (last_ref > ref_occ || last_seq > seq_occ) F.overlap = true;
            last_ref = ref_occ + mk.len; last_seq = seq_occ + mk.len;
        }
        const size_t ng = F.an.size() - 1;
        F.t_gap.assign(ng, -1); F.gap_score.assign(ng, 0); F.gap_cig.assign(ng, 0);
        if (!F.overlap) {
            last_ref = mem_pos + first.len; last_seq = (uint64_t)first.idx + first.len;
            for (size_t k = 1; k < F.an.size(); ++k) {
                const Mem& mk = R.mems[F.an[k].first];
                const Mem& mp = R.mems[F.an[k - 1].first];
                const uint64_t ref_occ = occ_of(R, F.an[k]), seq_occ = mk.idx;
                if (last_ref == ref_occ) {
                    if (last_seq < seq_occ) {                                          // pure insertion
                        const size_t l = seq_occ - last_seq;
                        F.gap_score[k - 1] = (int32_t)(-std::min((size_t)P.gapo + l * P.gape, (size_t)P.gapo2 + l * P.gape2));
                        F.gap_cig[k - 1] = (uint32_t)((l << 4) | 1);
                    }
                } else if (last_seq == seq_occ) {                                      // "deletion": l is computed as 0 (aligner_ksw2.hpp:2939)
                    const size_t l = seq_occ - last_seq;
                    F.gap_score[k - 1] = (int32_t)(-std::min((size_t)P.gapo + l * P.gape, (size_t)P.gapo2 + l * P.gape2));
                    F.gap_cig[k - 1] = (uint32_t)((l << 4) | 2);
                    F.t_gap[k - 1] = -2;                                               // marks "has a one-op CIGAR even if its length is 0"
                } else {
                    const uint64_t cc_occ = occ_of(R, F.an[k - 1]) + mp.len;
                    const uint64_t cc_len = ref_occ - cc_occ;
                    const uint64_t ccs_pos = (uint64_t)mp.idx + mp.len;
                    const uint64_t ccs_len = seq_occ - ccs_pos;
                    uint64_t q_off; int qmode;
                    qseg(ccs_pos, ccs_len, false, q_off, qmode);
                    F.t_gap[k - 1] = add(q_off, (int)ccs_len, qmode, cc_occ, (int)cc_len, 0, DP_EZ_RIGHT);
                }
                last_ref = ref_occ + mk.len; last_seq = seq_occ + mk.len;
            }
        }
    }

    // ---- fill_chain, part 2: extension results are in (aligner_ksw2.hpp:2852-2886, 2975-2996) ----
    // returns true if a dependent global problem was queued
    bool fill_after_ext(ReadState& R, const moni_dp_result_t* res, uint32_t base, std::vector<moni_dp_task_t>& tasks) {
        Fill& F = R.fill;
        const Mem& first = R.mems[F.an[0].first];
        const Mem& last = R.mems[F.an.back().first];
        int score_lc = 0, score_rc = 0;
        if (F.t_lc >= 0) { score_lc = res[base + F.t_lc].mqe; F.lc_mqe_t = res[base + F.t_lc].mqe_t; }
        if (F.t_rc >= 0) { score_rc = res[base + F.t_rc].mqe; F.rc_mqe_t = res[base + F.t_rc].mqe_t; }
        F.score.score = score_lc + score_rc;
        const uint64_t mem_pos = occ_of(R, F.an[0]);
        const uint64_t mem_len = occ_of(R, F.an.back()) + last.len - mem_pos;
        const uint64_t lq = (uint64_t)(int64_t)(F.lcs_len > 0 ? F.lc_mqe_t + 1 : 0);
        const uint64_t rq = (uint64_t)(int64_t)(F.rcs_len > 0 ? F.rc_mqe_t + 1 : 0);
        F.ref_pos = lq > mem_pos ? 0 : mem_pos - lq;
        F.ref_len = lq + mem_len + rq;
        F.score.pos = F.ref_pos;
        if (!F.overlap) {
            uint32_t sc = (uint32_t)F.score.score;
            for (size_t k = 1; k < F.an.size(); ++k) {
                const int32_t gs = F.t_gap[k - 1] >= 0 ? res[base + F.t_gap[k - 1]].score : F.gap_score[k - 1];
                sc += (uint32_t)((uint64_t)R.mems[F.an[k - 1].first].len * (uint64_t)P.smatch + (uint64_t)(int64_t)gs);
            }
            sc += (uint32_t)((uint64_t)last.len * (uint64_t)P.smatch);
            F.score.score = (int32_t)sc;
            return false;
        }
        // overlapping MEMs: one global alignment of the whole read against the window (aligner_ksw2.hpp:2984-2996, 3009-3015)
        moni_dp_task_t t;
        if (!F.strand) { t.q_off = R.off; t.reserved = DP_Q_READS | DP_T_TEXT; }
        else { t.q_off = R.off + R.m - 1; t.reserved = DP_Q_READS | DP_T_TEXT | DP_Q_REV | DP_Q_COMP; }
        t.qlen = (int32_t)R.m; t.t_off = F.ref_pos; t.tlen = (int32_t)F.ref_len;
        t.flag = F.score_only ? DP_EZ_SCORE_ONLY : DP_EZ_RIGHT;
        (void)first;
        F.t_glob = 0;                                    // the only task this read queues for the next round
        tasks.push_back(t);
        return true;
    }

    void fill_after_glob(ReadState& R, const moni_dp_result_t* res, uint32_t base) { R.fill.score.score = res[base + R.fill.t_glob].score; }

    void fill_validate(ReadState& R) {                                                  // aligner_ksw2.hpp:2998-2999
        Fill& F = R.fill;
        if (!ix.valid(F.ref_pos, F.ref_len)) F.score.score = INT32_MIN;
    }

    // ---- fill_chain, part 3 (final pass only): CIGAR, MD/NM, positions (aligner_ksw2.hpp:3000-3175) ----
    void fill_final(ReadState& R, const moni_dp_result_t* resA, uint32_t baseA, const std::vector<uint32_t>& cigA,
                    const moni_dp_result_t* resB, uint32_t baseB, const std::vector<uint32_t>& cigB) {
        Fill& F = R.fill;
        Sam& S = R.sam;
        if (!ix.valid(F.ref_pos, F.ref_len)) return;
        std::vector<uint32_t> cigar;
        auto slice = [](const moni_dp_result_t& r, const std::vector<uint32_t>& pool) { return std::make_pair(pool.data() + r.cigar_off, (size_t)r.n_cigar); };
        if (F.overlap) {
            auto c = slice(resB[baseB + F.t_glob], cigB);
            cigar.assign(c.first, c.first + c.second);
            F.score.score = resB[baseB + F.t_glob].score;
        } else {
            auto push_merge_first = [&](const uint32_t* c, size_t n) {                 // first op merges into a preceding M
                if (n > 0) { if ((c[0] & 0xf) == 0 && !cigar.empty()) cigar.back() += c[0]; else cigar.push_back(c[0]); }
                for (size_t k = 1; k < n; ++k) cigar.push_back(c[k]);
            };
            if (F.t_lc >= 0) { auto c = slice(resA[baseA + F.t_lc], cigA); for (size_t k = 0; k < c.second; ++k) cigar.push_back(c.first[c.second - 1 - k]); }
            for (size_t j = 0; j < F.an.size(); ++j) {
                const uint32_t mlen = R.mems[F.an[j].first].len;
                if (!cigar.empty() && (cigar.back() & 0xf) == 0) cigar.back() += mlen << 4;
                else cigar.push_back(mlen << 4);
                if (j + 1 < F.an.size()) {
                    if (F.t_gap[j] >= 0) { auto c = slice(resA[baseA + F.t_gap[j]], cigA); push_merge_first(c.first, c.second); }
                    else if (F.gap_cig[j] != 0 || F.t_gap[j] == -2) { const uint32_t op = F.gap_cig[j]; push_merge_first(&op, 1); }
                }
            }
            if (F.t_rc >= 0) { auto c = slice(resA[baseA + F.t_rc], cigA); push_merge_first(c.first, c.second); }
        }
        auto cig_string = [&]() { std::string s; for (uint32_t c : cigar) { s += std::to_string(c >> 4); s.push_back("MID"[c & 0xf]); } return s; };
        // nt4 views of the window and the read
        std::vector<uint8_t> ref(F.ref_len + 1), seq(R.m + 1);
        for (uint64_t k = 0; k < F.ref_len; ++k) ref[k] = nt4_of(F.ref_pos + k < ix.n_text ? ix.text[F.ref_pos + k] : 0);
        for (uint32_t k = 0; k < R.m; ++k) seq[k] = nt4_of(F.strand ? compl_of(reads[R.off + R.m - 1 - k]) : reads[R.off + k]);
        S.lift_cigar = cig_string();
        S.lift_md.clear();
        S.lift_nm = md_core(ref.data(), seq.data(), cigar, S.lift_md);
        const auto refi = ix.index(F.ref_pos);
        S.as = (size_t)(int64_t)F.score.score;
        S.lift_pos = refi.second + 1;
        S.lift_rname = ix.names[refi.first];
        const uint64_t lifted = ix.lift(F.ref_pos);                                     // aligner_ksw2.hpp:3133-3160
        std::vector<uint32_t> lcig;
        ix.lift_cigar_at(F.ref_pos, cigar.data(), (uint32_t)cigar.size(), lcig);
        const auto lft_ref = ix.index(lifted);
        S.pos = lft_ref.second + 1;
        S.rname = ix.names[lft_ref.first];
        S.cigar.clear();
        for (uint32_t c : lcig) { S.cigar += std::to_string(c >> 4); S.cigar.push_back("MID"[c & 0xf]); }
        uint64_t rl = 0;
        for (uint32_t c : lcig) { const int op = c & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += c >> 4; }
        if (rl > 0) {
            std::vector<uint8_t> lref(rl + 1);
            for (uint64_t k = 0; k < rl; ++k) lref[k] = nt4_of(lifted + k < ix.n_text ? ix.text[lifted + k] : 0);
            S.md.clear();
            S.nm = md_core(lref.data(), seq.data(), lcig, S.md);
            S.rlen = rl;
        } else {
            S.pos = 0; S.rname = "*"; S.cigar = "*"; S.rlen = 0; S.unmapped_lft = true;
        }
    }

    // aligner_ksw2.hpp:553-597
    bool check_left_mem(ReadState& R, size_t ci) {
        const Chain& ch = R.chains[ci];
        const uint32_t aid = ch.anchors.back();                                         // leftmost anchor (chains are stored right to left)
        const uint64_t left_pos = occ_of(R, R.anchors[aid]);
        const size_t left_ref = ix.index(ix.lift(left_pos)).second + 1;
        bool seen = false;
        for (auto& lv : R.left_mem_vec) {
            const size_t d = lv.first > left_ref ? lv.first - left_ref : left_ref - lv.first;
            if (d < P.region_dist && lv.second == (size_t)ch.score) seen = true;
        }
        if (seen) return true;
        R.left_mem_vec.push_back(std::make_pair(left_ref, (size_t)ch.score));
        return false;
    }

    // a scored chain comes back into the selection loop (aligner_ksw2.hpp:436-460, 528-548)
    void absorb_score(ReadState& R, Score score) {
        score.lft = ix.lift(score.pos);
        if (score.score > R.max_score) { R.max_score = score.score; R.sam.alt_haplotypes.clear(); R.sam.alt_pos.clear(); R.sam.alt_scores.clear(); }
        else if (score.score == R.max_score) {
            const auto ref = ix.index(score.pos);
            R.sam.alt_haplotypes.push_back(ix.names[ref.first]); R.sam.alt_pos.push_back(ref.second + 1); R.sam.alt_scores.push_back((size_t)(int64_t)score.score);
        }
        bool replaced = false;
        size_t& i = R.i;
        for (size_t j = 0; j < R.best_scores.size(); ++j) {
            const size_t bl = std::get<1>(R.best_scores[j]);
            const size_t d = bl > score.lft ? bl - score.lft : score.lft - bl;
            if (d < P.region_dist) {
                if (score.score > std::get<0>(R.best_scores[j])) {
                    if (replaced) R.best_scores[j] = std::make_tuple(0, (size_t)0, i - 1);
                    else { R.best_scores[j] = std::make_tuple(score.score, (size_t)score.lft, i); i++; replaced = true; }
                } else {
                    j = R.best_scores.size(); replaced = true; i++;
                }
            }
        }
        if (!replaced) { R.best_scores.push_back(std::make_tuple(score.score, (size_t)score.lft, i)); i++; }
    }


    // Runs the read until it needs DP results (tasks appended) or is done.
    void advance(ReadState& R, std::vector<moni_dp_task_t>& tasks) {
        while (R.stage == ReadState::LOOP) {
            if (R.i < R.chains.size() && R.different_scores.size() < P.check_k) {
                { const size_t v = (size_t)R.chains[R.i].score; bool f = false; for (size_t q = 0; q < R.different_scores.size(); ++q) f = f || R.different_scores[q] == v;
                  if (!f) { const size_t q = R.different_scores.size(); R.different_scores.resize(q + 1); R.different_scores[q] = v; } }
                if (P.left_mem_check && check_left_mem(R, R.i)) { ++R.i; ++R.csv.num_chains_skipped; continue; }      // (aligner_ksw2.hpp:417)
                if (R.different_scores.size() < P.check_k) {
                    fill_begin(R, R.chains[R.i], true, tasks);
                    R.stage = ReadState::WAIT_A;
                    return;
                }
                continue;   // the while condition of the reference fails next time round
            }
            // after the loop (aligner_ksw2.hpp:464-509)
            while (R.best_scores.size() < 2) R.best_scores.push_back(std::make_tuple(0, (size_t)0, R.chains.size()));
            std::sort(R.best_scores.begin(), R.best_scores.end(), std::greater<std::tuple<int32_t, size_t, size_t>>());
            if (std::get<0>(R.best_scores[0]) < R.min_score) { R.stage = ReadState::DONE; return; }
            R.score2 = std::get<0>(R.best_scores[1]);
            R.final_chain = std::get<2>(R.best_scores[0]);
            if (R.final_chain >= R.chains.size() || R.chain_score_cache[R.final_chain] < R.min_score) { R.stage = ReadState::DONE; return; }
            // chain_score(..., score_only = false): the score-only pass was already done for this chain in the loop;
            // its (cached) score is >= min_score here, so the final pass always runs (aligner_ksw2.hpp:2062-2065)
            fill_begin(R, R.chains[R.final_chain], false, tasks);
            R.stage = ReadState::FINAL_WAIT_A;
            return;
        }
    }

    // Host part of an alignment that was decided on the GPU (align_core.h): positions, MD/NM, MAPQ
    // (aligner_ksw2.hpp:3110-3175, 2067-2076, 498-511).
    void finish_record(uint32_t m, uint64_t off, uint32_t strand, uint64_t ref_pos, int32_t score, int32_t score2, const uint32_t* cig,
                       uint32_t n_cig, const uint64_t* alt_pos, const int32_t* alt_score, uint32_t n_alt, Sam& S) const {
        std::vector<uint32_t> cigar(cig, cig + n_cig);
        uint64_t ref_len = 0;
        for (uint32_t c : cigar) { const int op = c & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += c >> 4; }
        std::string cs;
        for (uint32_t c : cigar) { cs += std::to_string(c >> 4); cs.push_back("MID"[c & 0xf]); }
        std::vector<uint8_t> ref(ref_len + 1), seq(m + 1);
        for (uint64_t k = 0; k < ref_len; ++k) ref[k] = nt4_of(ref_pos + k < ix.n_text ? ix.text[ref_pos + k] : 0);
        for (uint32_t k = 0; k < m; ++k) seq[k] = nt4_of(strand ? compl_of(reads[off + m - 1 - k]) : reads[off + k]);
        S.lift_cigar = cs;
        S.lift_md.clear();
        S.lift_nm = md_core(ref.data(), seq.data(), cigar, S.lift_md);      // the window the CIGAR spans == [ref_pos, ref_pos + rlen)
        const auto refi = ix.index(ref_pos);
        S.as = (size_t)(int64_t)score;
        S.lift_pos = refi.second + 1;
        S.lift_rname = ix.names[refi.first];
        const uint64_t lifted = ix.lift(ref_pos);                              // aligner_ksw2.hpp:3133-3160
        std::vector<uint32_t> lcig;
        ix.lift_cigar_at(ref_pos, cig, n_cig, lcig);
        const auto lft_ref = ix.index(lifted);
        S.pos = lft_ref.second + 1; S.rname = ix.names[lft_ref.first];
        S.cigar.clear();
        for (uint32_t c : lcig) { S.cigar += std::to_string(c >> 4); S.cigar.push_back("MID"[c & 0xf]); }
        uint64_t rl = 0;
        for (uint32_t c : lcig) { const int op = c & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += c >> 4; }
        if (rl > 0) {
            std::vector<uint8_t> lref(rl + 1);
            for (uint64_t k = 0; k < rl; ++k) lref[k] = nt4_of(lifted + k < ix.n_text ? ix.text[lifted + k] : 0);
            S.md.clear();
            S.nm = md_core(lref.data(), seq.data(), lcig, S.md);
            S.rlen = rl;
        }
        else { S.pos = 0; S.rname = "*"; S.cigar = "*"; S.rlen = 0; S.unmapped_lft = true; }
        for (uint32_t k = 0; k < n_alt; ++k) {
            const auto r = ix.index(alt_pos[k]);
            S.alt_haplotypes.push_back(ix.names[r.first]); S.alt_pos.push_back(r.second + 1); S.alt_scores.push_back((size_t)(int64_t)alt_score[k]);
        }
        S.flag = strand ? 16 : 0;
        S.zs = (size_t)(int64_t)score2;
        S.mapq = mapq_se_bwa((int32_t)S.as, (int32_t)S.zs, (int32_t)S.rlen, (int32_t)m, (int32_t)P.min_len, P.smatch, P.smismatch, 50.0, mapq_coeff_fac);
        if (strand) S.rev_read = true;
    }

    // finish_record + sam_write in one pass straight into the output text, no per-read containers: what the host stage
    // of the align-kernel path runs per read (same bytes as the two functions above and below produce; sam.hpp:144-188,249-287).
    struct OutBuf {                    // grow-only text buffer (kept by the caller across batches: its pages stay mapped)
        char* base = nullptr; size_t len = 0, cap = 0;
        bool ensure(size_t extra) {
            if (len + extra <= cap) return true;
            size_t nc = cap ? cap + cap / 2 : (size_t)1 << 20;
            while (nc < len + extra) nc += nc / 2;
            char* nb = (char*)realloc(base, nc);
            if (!nb) return false;
            base = nb; cap = nc; return true;
        }
        void release() { free(base); base = nullptr; len = cap = 0; }
    };
    static inline char* put_int(char* p, int v) {
        char b[12]; int n = 0; unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
        do { b[n++] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) *p++ = '-';
        while (n) *p++ = b[--n];
        return p;
    }
#define PUT_LIT(p, lit) put_str(p, lit, sizeof(lit) - 1)
    static inline char* put_str(char* p, const char* s, size_t n) { memcpy(p, s, n); return p + n; }
    static inline char* put_str(char* p, const std::string& s) { memcpy(p, s.data(), s.size()); return p + s.size(); }
    // upper bound of one record's text
    size_t record_bound(size_t name_len, uint32_t m, uint32_t n_cig, uint32_t n_alt) const {
        return name_len + 2 * (size_t)m + 22 * (size_t)n_cig + 4 * (size_t)m + 64 * (size_t)n_cig + (size_t)n_alt * (max_name_len + 26) + 2 * max_name_len + 256;
    }
    bool emit_record(OutBuf& ob, std::vector<char>& md_scratch, const char* name, size_t name_len, const uint8_t* rd, const uint8_t* ql, uint32_t m,
                     bool aligned, uint32_t strand, uint64_t ref_pos, int32_t score, int32_t score2, const uint32_t* cig, uint32_t n_cig,
                     const moni_alt_like* alts, uint32_t n_alt, const char* md_given = nullptr, uint32_t md_given_len = 0, int nm_given = 0,
                     int lift_nm_given = 0) const {
        // cig / ref_pos: the alignment on the pangenome text (the OA tag); columns 3, 4, 6 and MD / NM are its lift (aligner_ksw2.hpp:3133-3160).
        // md_given: MD:Z text and NM of the lifted alignment and NM of the unlifted one, computed by the align kernel; else computed here
        if (!aligned) {
            if (!ob.ensure(name_len + 2 * (size_t)m + 64)) return false;
            char* p = ob.base + ob.len;
            p = put_str(p, name, name_len);
            p = PUT_LIT(p, "\t4\t*\t0\t255\t*\t*\t0\t0\t");
            p = put_str(p, (const char*)rd, m); *p++ = '\t';
            if (ql) p = put_str(p, (const char*)ql, m); else *p++ = '*';
            *p++ = '\n';
            ob.len = (size_t)(p - ob.base);
            return true;
        }
        static thread_local std::vector<uint32_t> lcig;
        ix.lift_cigar_at(ref_pos, cig, n_cig, lcig);
        const uint32_t n_lcig = (uint32_t)lcig.size();
        const uint64_t lifted = ix.lift(ref_pos);
        uint64_t ref_len = 0, del_len = 0;
        for (uint32_t k = 0; k < n_lcig; ++k) { const uint32_t c = lcig[k]; const int op = c & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += c >> 4; if (op == 2) del_len += c >> 4; }
        if (!ob.ensure(record_bound(name_len, m, n_cig + n_lcig, n_alt) + 2 * del_len)) return false;
        char* p = ob.base + ob.len;
        p = put_str(p, name, name_len);
        // MD / NM over the window a CIGAR spans (write_MD_core); NM is printed before MD, so MD goes through a scratch
        { const size_t need = 3 * (size_t)m + 2 * del_len + 40 * (size_t)(n_cig + n_lcig) + 64 + (md_given ? (size_t)md_given_len : 0); if (md_scratch.size() < need) md_scratch.resize(need); }
        char* const md0 = md_scratch.data(); char* md = md0;
        int NM = 0, NM_unlifted = 0;
        if (md_given) {
            memcpy(md, md_given, md_given_len);
            md += md_given_len;
            NM = nm_given; NM_unlifted = lift_nm_given;
        } else {
            auto tb = [&](uint64_t a) -> uint8_t { return nt4_of(a < ix.n_text ? ix.text[a] : 0); };
            auto qb = [&](uint32_t k) -> uint8_t { return nt4_of(strand ? compl_of(rd[m - 1 - k]) : rd[k]); };
            auto pass = [&](const uint32_t* cg, uint32_t ncg, uint64_t t, char* out) -> int {          // out == nullptr: count only
                int l_MD = 0, nm = 0; uint32_t q = 0;
                for (uint32_t i = 0; i < ncg; ++i) {
                    const int op = cg[i] & 0xf, len = (int)(cg[i] >> 4);
                    if (op == 0 || op == 7 || op == 8) {
                        for (int j = 0; j < len; ++j) {
                            const uint8_t tc = tb(t + j);
                            if (qb(q + j) != tc) { if (out) { out = put_int(out, l_MD); *out++ = "ACGTN"[tc]; } l_MD = 0; ++nm; }
                            else ++l_MD;
                        }
                        q += len; t += len;
                    } else if (op == 1) { q += len; nm += len; }
                    else if (op == 2) {
                        if (out) { out = put_int(out, l_MD); *out++ = '^'; for (int j = 0; j < len; ++j) *out++ = "ACGTN"[tb(t + j)]; }
                        l_MD = 0; t += len; nm += len;
                    } else if (op == 3) t += len;
                }
                if (out) { if (l_MD > 0) out = put_int(out, l_MD); md = out; }
                return nm;
            };
            NM_unlifted = pass(cig, n_cig, ref_pos, nullptr);
            if (ref_len > 0) NM = pass(lcig.data(), n_lcig, lifted, md);
        }
        const size_t md_len = (size_t)(md - md0);
        char* const cs0 = md;                         // lifted CIGAR text, then the unlifted one, behind the MD text in the scratch
        for (uint32_t k = 0; k < n_lcig; ++k) { md = put_int(md, (int)(lcig[k] >> 4)); *md++ = "MID"[lcig[k] & 0xf]; }
        const size_t cs_len = (size_t)(md - cs0);
        char* const os0 = md;
        for (uint32_t k = 0; k < n_cig; ++k) { md = put_int(md, (int)(cig[k] >> 4)); *md++ = "MID"[cig[k] & 0xf]; }
        const size_t os_len = (size_t)(md - os0);
        const auto refi = ix.index(ref_pos);
        const std::string& lift_rname = ix.names[refi.first];
        const int lift_pos_1 = (int)(refi.second + 1);
        const auto lfti = ix.index(lifted);
        const bool mapped = ref_len > 0;              // else: pos 0, rname/cigar "*", no MD, NM 0, tags still printed (unmapped_lft)
        const int flag = strand ? 16 : 0;
        const int mapq = (int)mapq_se_bwa(score, score2, (int32_t)(mapped ? ref_len : 0), (int32_t)m, (int32_t)P.min_len, P.smatch, P.smismatch, 50.0, mapq_coeff_fac);
        *p++ = '\t'; p = put_int(p, flag); *p++ = '\t';
        if (mapped) p = put_str(p, ix.names[lfti.first]); else *p++ = '*';
        *p++ = '\t'; p = put_int(p, mapped ? (int)(lfti.second + 1) : 0); *p++ = '\t'; p = put_int(p, mapq); *p++ = '\t';
        if (mapped) p = put_str(p, cs0, cs_len); else *p++ = '*';
        p = PUT_LIT(p, "\t*\t0\t0\t");
        if (strand) { for (uint32_t k = 0; k < m; ++k) p[k] = (char)compl_of(rd[m - 1 - k]); p += m; }
        else p = put_str(p, (const char*)rd, m);
        *p++ = '\t';
        if (ql) { if (strand) { for (uint32_t k = 0; k < m; ++k) p[k] = (char)ql[m - 1 - k]; p += m; } else p = put_str(p, (const char*)ql, m); }
        else *p++ = '*';
        p = PUT_LIT(p, "\tAS:i:"); p = put_int(p, score); p = PUT_LIT(p, "\tNM:i:"); p = put_int(p, mapped ? NM : 0);
        if (score2 != 0) { p = PUT_LIT(p, "\tZS:i:"); p = put_int(p, score2); }
        p = PUT_LIT(p, "\tMD:Z:"); if (mapped) p = put_str(p, md0, md_len);
        p = PUT_LIT(p, "\tOA:Z:"); p = put_str(p, lift_rname); *p++ = ','; p = put_int(p, lift_pos_1);
        p = put_str(p, strand ? ",-," : ",+,", 3); p = put_str(p, os0, os_len); *p++ = ','; p = put_int(p, mapq); *p++ = ','; p = put_int(p, NM_unlifted); *p++ = ';';
        p = PUT_LIT(p, "\tAA:Z:");
        for (uint32_t k = 0; k < n_alt; ++k) {
            const auto r = ix.index(alts[k].pos);
            p = put_str(p, ix.names[r.first]); *p++ = ','; p = put_int(p, (int)(r.second + 1)); *p++ = ','; p = put_int(p, alts[k].score); *p++ = ';';
        }
        *p++ = '\n';
        ob.len = (size_t)(p - ob.base);
        return true;
    }

    static void sam_write(std::string& out, const Sam& s, const std::string& name, const std::string& seq, const std::string* qual) {   // sam.hpp:144-188
        char buf[32];
        auto d = [&](size_t v) { snprintf(buf, sizeof buf, "%d", (int)v); out += buf; };
        out += name; out.push_back('\t'); d(s.flag); out.push_back('\t'); out += s.rname; out.push_back('\t'); d(s.pos); out.push_back('\t');
        d(s.mapq); out.push_back('\t'); out += s.cigar; out += "\t*\t0\t0\t"; out += seq; out.push_back('\t');
        if (qual) out += *qual; else out.push_back('*');
        if (!(s.flag & 4) || s.unmapped_lft) {
            out += "\tAS:i:"; d(s.as); out += "\tNM:i:"; d(s.nm);
            if (s.zs > 0) { out += "\tZS:i:"; d(s.zs); }
            out += "\tMD:Z:"; out += s.md; out += "\tOA:Z:"; out += s.lift_rname; out.push_back(','); d(s.lift_pos);
            out += (s.flag & 16) ? ",-," : ",+,"; out += s.lift_cigar; out.push_back(','); d(s.mapq); out.push_back(','); d(s.lift_nm); out.push_back(';');
            out += "\tAA:Z:";
            for (size_t i = 0; i < s.alt_haplotypes.size(); ++i) { out += s.alt_haplotypes[i]; out.push_back(','); d(s.alt_pos[i]); out.push_back(','); d(s.alt_scores[i]); out.push_back(';'); }
        }
        out.push_back('\n');
    }
};

// Persistent worker pool for the host stages: a batch goes through ~100 short parallel phases, far too many to pay a
// thread spawn each time.
class Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::function<void(int)> job;
    uint64_t gen = 0;
    int pending = 0;
    bool quit = false;
public:
    const int n;
    explicit Pool(int n_) : n(n_ < 1 ? 1 : n_) {
        for (int t = 1; t < n; ++t) th.emplace_back([this, t]() {
            uint64_t seen = 0;
            while (true) {
                std::function<void(int)> j;
                { std::unique_lock<std::mutex> lk(mu); cv_go.wait(lk, [&] { return quit || gen != seen; }); if (quit) return; seen = gen; j = job; }
                j(t);
                { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) cv_done.notify_one(); }
            }
        });
    }
    ~Pool() { { std::lock_guard<std::mutex> lk(mu); quit = true; } cv_go.notify_all(); for (auto& x : th) x.join(); }
    void run(const std::function<void(int)>& fn) {
        if (n == 1) { fn(0); return; }
        { std::lock_guard<std::mutex> lk(mu); job = fn; pending = n - 1; ++gen; }
        cv_go.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};

template <class Fn>
static void parallel_for(Pool& pool, size_t n, Fn fn) {
    if (pool.n <= 1 || n < 2) { fn(0, (size_t)0, n); return; }
    const int T = pool.n;
    pool.run([&](int t) { const size_t lo = n * t / T, hi = n * (t + 1) / T; if (lo < hi) fn(t, lo, hi); });
}

struct AlignStats { uint64_t reads = 0, aligned = 0, dp_tasks = 0, dp_cells = 0, dp_rounds = 0, handed_back = 0, dp_reused = 0, dp_cells_reused = 0, kernel_fallback = 0, dp_ref_bytes = 0, dp_cells_cut = 0, dp_slots = 0; double t_k_chain = 0, t_k_dp = 0, t_k_select = 0, t_k_finish = 0; double t_seed = 0, t_chain = 0, t_dp = 0, t_host = 0; };

static inline double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

// The whole batch: seeds -> SAM records (no header), in read order.
static int align_batch(Backend& be, const HostIndex& ix, const moni_align_params_t& P, const uint8_t* reads, const uint64_t* offs,
                       uint64_t n_reads, const uint8_t* names, const uint64_t* name_off, const uint8_t* quals, std::string& sam_out,
                       AlignStats& st, const std::function<int(std::vector<uint64_t>&)>* genome_hi_lo = nullptr, std::string* csv_out = nullptr) {
    // csv_out (`-c`): one line of MEM statistics per read (write_csv, csv.hpp:55-67); genome_hi_lo fetches, for every seed of the batch, the largest and
    // the smallest count of its occurrences on any one genome (largest | smallest << 32; a kernel: seed_core.h genome_task)
    const int T = P.host_threads > 0 ? (int)P.host_threads : 1;
    Pool pool(T);
    std::vector<moni_mem_t> gm;
    std::vector<uint64_t> go, rmo;
    moni_seed_params_t sp;
    sp.min_len = P.min_len; sp.filter_seeds = P.filter_seeds; sp.n_seeds_thr = P.n_seeds_thr; sp.report_mems = 0;
    double t0 = now_s();
    int rc = be.seed(sp, gm, go, rmo);
    if (rc) return rc;
    std::vector<uint64_t> hi_lo;
    if (csv_out && genome_hi_lo && (rc = (*genome_hi_lo)(hi_lo))) return rc;
    if (csv_out && hi_lo.size() != gm.size()) return MONI_EINVAL;
    st.t_seed += now_s() - t0;
    t0 = now_s();
    Aligner A(ix, P, reads, offs);
    std::vector<ReadState> RS(n_reads);
    // frequency filter + chaining (aligner_ksw2.hpp:342, 382, 394)
    parallel_for(pool, n_reads, [&](int, size_t lo, size_t hi) {
        for (size_t r = lo; r < hi; ++r) {
            ReadState& R = RS[r];
            R.off = offs[r] - offs[0]; R.m = (uint32_t)(offs[r + 1] - offs[r]);
            const uint64_t a = rmo[r], b = rmo[r + 1];
            R.mems.reserve(b - a);
            for (uint64_t k = a; k < b; ++k) { const moni_mem_t& g = gm[k]; R.mems.push_back(Mem{g.pos, g.len, g.idx, g.rpos, g.mate, go.data() + g.occ_off, g.occ_cnt}); }
            if (csv_out) {                                                                // calculate_MEM_stats (aligner_ksw2.hpp:1868-1902)
                CsvRead& C = R.csv;
                C.num_uniq_mems = b - a;
                for (uint64_t k = a; k < b; ++k) { C.total_mem_occ += gm[k].total_occ; C.num_mems_filter += gm[k].num_filtered; }
                for (uint64_t k = a; k < b; ++k) {
                    const double mem_freq = (gm[k].occ_cnt / (static_cast<double>(C.total_mem_occ)));
                    C.max_mem_freq = (C.max_mem_freq > mem_freq ? C.max_mem_freq : mem_freq);
                    C.min_mem_freq = (C.min_mem_freq > mem_freq ? mem_freq : C.min_mem_freq);
                    const size_t hi = (size_t)(hi_lo[k] & 0xFFFFFFFFull), lo = (size_t)(hi_lo[k] >> 32);      // over the seed's count_dict; every count is >= 1
                    if (hi) {
                        if (C.high_occ_mem == 0 && C.low_occ_mem == 0) { C.high_occ_mem = hi; C.low_occ_mem = lo; }
                        else { C.high_occ_mem = (C.high_occ_mem > hi ? C.high_occ_mem : hi); C.low_occ_mem = (C.low_occ_mem > lo ? lo : C.low_occ_mem); }
                    }
                }
            }
            size_t total = 0;
            for (auto& m : R.mems) total += m.nocc;
            if (P.filter_freq) {                                                          // seed_freq_filter
                std::vector<Mem> keep;
                for (auto& m : R.mems) { const double fr = static_cast<double>(m.nocc) / total; if (!(fr > P.freq_thr)) keep.push_back(m); else R.csv.num_mems_filter += m.nocc; }
                R.mems.swap(keep);
            }
            size_t na = 0;
            for (auto& m : R.mems) na += m.nocc;
            const bool chained = na > 0 && chain_mems(R.mems, R.anchors, R.chains, P);
            if (!chained) { R.stage = ReadState::DONE; continue; }
            R.min_score = (int32_t)(20 + 8 * log((double)R.m));
            R.chain_score_cache.assign(R.chains.size(), INT32_MIN);
        }
    });
    st.t_chain += now_s() - t0;
    moni_dp_params_t dp;
    memset(&dp, 0, sizeof dp);
    dp.m = 5;
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) dp.mat[i * 5 + j] = i == j ? P.smatch : (int8_t)-P.smismatch; dp.mat[i * 5 + 4] = 0; }
    for (int j = 0; j < 5; ++j) dp.mat[20 + j] = 0;
    dp.q = P.gapo; dp.e = P.gape; dp.w = P.w; dp.zdrop = P.zdrop; dp.end_bonus = P.end_bonus;

    std::vector<std::vector<moni_dp_task_t>> ttasks(T);
    std::vector<std::vector<uint32_t>> temit(T);           // reads that queued tasks, per thread
    std::vector<moni_dp_task_t> tasks;
    std::vector<moni_dp_result_t> res;
    std::vector<uint32_t> cig;
    std::vector<uint32_t> waiting;

    auto finish_final = [&](ReadState& R) {                // tail of chain_score + align (aligner_ksw2.hpp:2067-2076, 498-511)
        Sam& S = R.sam;
        S.flag = R.fill.strand ? 16 : 0;
        S.zs = (size_t)(int64_t)R.score2;
        S.mapq = mapq_se_bwa((int32_t)S.as, (int32_t)S.zs, (int32_t)S.rlen, (int32_t)R.m, (int32_t)P.min_len, P.smatch, P.smismatch, 50.0, A.mapq_coeff_fac);
        if (R.fill.strand) { S.rev_read = true; S.flag |= 16; }
        R.aligned = true;
        R.stage = ReadState::DONE;
    };
    // Drive one read until it has queued DP problems for the next batch or is done.  `res`/`cig` are the results of the
    // batch the read was waiting on (unused on the first call).  A fill_chain that needs no DP at all (the MEM chain
    // covers the whole read) falls straight through.
    auto drive = [&](ReadState& R, std::vector<moni_dp_task_t>& q) {
        const uint32_t base = R.task_base;
        while (true) {
            const size_t qn = q.size();
            switch (R.stage) {
                case ReadState::LOOP:
                    A.advance(R, q);
                    if (R.stage == ReadState::DONE || q.size() != qn) return;
                    break;                                   // a fill with no DP problems: continue as if results had arrived
                case ReadState::WAIT_A:
                case ReadState::WAIT_B:
                    if (R.stage == ReadState::WAIT_A) {
                        if (A.fill_after_ext(R, res.data(), base, q)) { R.stage = ReadState::WAIT_B; return; }
                    } else {
                        A.fill_after_glob(R, res.data(), base);
                    }
                    A.fill_validate(R);
                    R.chain_score_cache[R.i] = R.fill.score.score;
                    A.absorb_score(R, R.fill.score);
                    R.stage = ReadState::LOOP;
                    break;
                case ReadState::FINAL_WAIT_A:
                    if (A.fill_after_ext(R, res.data(), base, q)) { R.stage = ReadState::FINAL_WAIT_B; return; }
                    A.fill_final(R, res.data(), base, cig, nullptr, 0, cig);
                    finish_final(R);
                    return;
                case ReadState::FINAL_WAIT_B:
                    A.fill_final(R, nullptr, 0, cig, res.data(), base, cig);
                    finish_final(R);
                    return;
                default:
                    return;
            }
        }
    };
    auto with_queue = [&](int t, uint32_t r) {
        ReadState& R = RS[r];
        const size_t before = ttasks[t].size();
        drive(R, ttasks[t]);
        if (ttasks[t].size() != before) { R.task_base = (uint32_t)before; R.owner_thread = t; temit[t].push_back(r); }
    };
    double tt_drive = 0, tt_merge = 0, tt_sam = 0; double tq = now_s();
    // round 0: every chained read runs until its first DP request
    {
        std::vector<uint32_t> act;
        for (size_t r = 0; r < n_reads; ++r) if (RS[r].stage != ReadState::DONE) act.push_back((uint32_t)r);
        parallel_for(pool, act.size(), [&](int t, size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) with_queue(t, act[k]); });
    }
    tt_drive += now_s() - tq;
    while (true) {
        // merge the per-thread queues into one batch
        tq = now_s();
        tasks.clear(); waiting.clear();
        std::vector<uint32_t> tb(T + 1, 0);
        for (int t = 0; t < T; ++t) {
            tb[t + 1] = tb[t] + (uint32_t)ttasks[t].size();
            tasks.insert(tasks.end(), ttasks[t].begin(), ttasks[t].end());
            for (uint32_t r : temit[t]) { RS[r].task_base += tb[t]; waiting.push_back(r); }
            ttasks[t].clear(); temit[t].clear();
        }
        if (tasks.empty()) break;
        st.dp_tasks += tasks.size(); st.dp_rounds++;
        tt_merge += now_s() - tq;
        for (auto& t : tasks) st.dp_cells += (uint64_t)(t.qlen > 0 ? t.qlen : 0) * (uint64_t)(t.tlen > 0 ? t.tlen : 0);
        st.t_host += now_s() - t0;
        t0 = now_s();
        res.resize(tasks.size());
        rc = be.dp(dp, tasks, res, cig);
        if (rc) return rc;
        st.t_dp += now_s() - t0;
        t0 = now_s();
        // consume results; reads that continue queue their next problems for the next batch
        tq = now_s();
        parallel_for(pool, waiting.size(), [&](int t, size_t lo, size_t hi) { for (size_t k = lo; k < hi; ++k) with_queue(t, waiting[k]); });
        tt_drive += now_s() - tq;
    }
    tq = now_s();
    // SAM text in read order (align_reads_dispatcher.hpp:346-357, sam.hpp:144-188)
    std::vector<std::string> parts(T);
    parallel_for(pool, n_reads, [&](int t, size_t lo, size_t hi) {
        std::string& out = parts[t];
        std::string seq, qual, name;
        for (size_t r = lo; r < hi; ++r) {
            const ReadState& R = RS[r];
            const uint8_t* s = reads + R.off;
            name.assign((const char*)names + name_off[r], (const char*)names + name_off[r + 1]);
            seq.resize(R.m);
            if (R.sam.rev_read) for (uint32_t k = 0; k < R.m; ++k) seq[k] = (char)compl_of(s[R.m - 1 - k]);
            else seq.assign((const char*)s, (const char*)s + R.m);
            if (quals) {
                const uint8_t* qv = quals + R.off;
                qual.resize(R.m);
                if (R.sam.rev_read) for (uint32_t k = 0; k < R.m; ++k) qual[k] = (char)qv[R.m - 1 - k];
                else qual.assign((const char*)qv, (const char*)qv + R.m);
            }
            Sam S = R.sam;
            if (!R.aligned) S.flag = 4;                                                   // set_sam_not_aligned
            Aligner::sam_write(out, S, name, seq, quals ? &qual : nullptr);
        }
    });
    sam_out.clear();
    for (auto& p : parts) sam_out += p;
    if (csv_out) {                                                                         // write_csv (csv.hpp:55-67), read order
        csv_out->clear();
        char buf[256];
        for (size_t r = 0; r < n_reads; ++r) {
            const CsvRead& C = RS[r].csv;
            csv_out->append((const char*)names + name_off[r], (size_t)(name_off[r + 1] - name_off[r]));
            snprintf(buf, sizeof buf, ",%zu,%zu,%f,%f,%zu,%zu,%zu,%zu\n", C.num_uniq_mems, C.total_mem_occ, C.max_mem_freq, C.min_mem_freq, C.high_occ_mem, C.low_occ_mem,
                     C.num_mems_filter, C.num_chains_skipped);
            *csv_out += buf;
        }
    }
    tt_sam += now_s() - tq;
    if (getenv("MH_TIMES")) fprintf(stderr, "align_host: drive %.3f merge %.3f sam %.3f s\n", tt_drive, tt_merge, tt_sam);
    st.reads += n_reads;
    for (auto& R : RS) if (R.aligned) st.aligned++;
    st.t_host += now_s() - t0;
    return MONI_OK;
}

}  // namespace mh
