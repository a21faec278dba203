// ORACLE — TEST INFRASTRUCTURE ONLY (see flat_index.hpp header).  PARITY UNPINNED.
//
// align.hpp: the single-end `moni align` per-read path after seeding, restated from
//   include/aligner/chain.hpp:72-438          find_chains (minimap2-style anchor chaining)
//   include/aligner/aligner_ksw2.hpp:328-521  aligner::align (SE): filters, chain selection, second-best score
//   include/aligner/aligner_ksw2.hpp:528-597  check_max_score / check_left_MEM
//   include/aligner/aligner_ksw2.hpp:1905-1933 seed_freq_filter
//   include/aligner/aligner_ksw2.hpp:2018-2098 chain_score
//   include/aligner/aligner_ksw2.hpp:2752-3196 fill_chain (every ksw2 call, CIGAR stitching, MD/NM, lift)
//   include/aligner/mapq.hpp:146-184          compute_mapq_se_bwa
//   include/common/sam.hpp:47-188,249-287     sam_t, write_sam, write_MD_core
// Deliberately literal about the reference's arithmetic quirks (unsigned wrap, `%d` of size_t, the
// deletion shortcut that computes l = 0, the left-context length when mem_pos <= ext_len, unstable
// std::sort on the same initial order with the same comparators, libstdc++).
// Lift-over (aligner_ksw2.hpp:3133-3175, liftidx.hpp:89-95,159-164): FlatIndex::lift / lift_cigar restate levioSAM's
// lift_pos (pinned by the reference's .ldx/.lft fixture) and lift_cigar ([UPSTREAM-RECALL], unpinned).
#pragma once
#include <climits>
#include <cmath>
#include <set>
#include <tuple>

#include "flat_index.hpp"
#include "ksw2.hpp"
#include "seed.hpp"

namespace oracle {

typedef long long int ll;

// ---- common.hpp:533-550 ------------------------------------------------------------------------------
static const char LogTable256[256] = {
#define LT(n) n, n, n, n, n, n, n, n, n, n, n, n, n, n, n, n
    -1, 0, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, LT(4), LT(5), LT(5), LT(6), LT(6), LT(6), LT(6),
    LT(7), LT(7), LT(7), LT(7), LT(7), LT(7), LT(7), LT(7)
#undef LT
};
static inline int ilog2_32(uint32_t v) {
    uint32_t t, tt;
    if ((tt = v >> 16)) return (t = tt >> 8) ? 24 + LogTable256[t] : 16 + LogTable256[tt];
    return (t = v >> 8) ? 8 + LogTable256[t] : LogTable256[v];
}
#define ORC_DIST(a, b) ((a) > (b) ? ((a) - (b)) : ((b) - (a)))

// ---- chain.hpp ------------------------------------------------------------------------------------------
struct chain_t {   // chain.hpp:25-52
    ll score = 0;
    size_t mate = 2;
    bool paired = false;
    bool reversed = false;
    std::vector<size_t> anchors;
    void reverse() { if (!reversed) { std::reverse(anchors.begin(), anchors.end()); reversed = true; } }
    void reset() { if (reversed) { std::reverse(anchors.begin(), anchors.end()); reversed = false; } }
};

struct chain_config_t {   // chain.hpp:72-80
    ll G = LLONG_MAX;
    ll max_dist_x = 500, max_dist_y = 100, max_iter = 10, max_pred = 5, min_chain_score = 40, min_chain_length = 1;
};

// chain.hpp:221-438
inline bool find_chains(const std::vector<mem_t>& mems, std::vector<std::pair<size_t, size_t>>& anchors,
                        std::vector<chain_t>& chains, const chain_config_t config = chain_config_t()) {
    auto cmp = [&](const std::pair<size_t, size_t>& i, const std::pair<size_t, size_t>& j) -> bool {
        return (mems[i.first].occs[i.second] + mems[i.first].len - 1) < (mems[j.first].occs[j.second] + mems[j.first].len - 1);
    };
    size_t tot_mem_length = 0;
    for (size_t i = 0; i < mems.size(); ++i) {             // populate_anchors, chain.hpp:83-95
        for (size_t j = 0; j < mems[i].occs.size(); ++j) anchors.push_back(std::make_pair(i, j));
        tot_mem_length += mems[i].len * mems[i].occs.size();
    }
    float avg_mem_length = (float)tot_mem_length / anchors.size();
    std::sort(anchors.begin(), anchors.end(), cmp);

    const ll G = config.G;
    const ll max_dist_x = config.max_dist_x, max_dist_y = config.max_dist_y, max_iter = config.max_iter;
    const ll max_pred = config.max_pred, min_chain_score = config.min_chain_score, min_chain_length = config.min_chain_length;

    std::vector<ll> f(anchors.size(), 0), p(anchors.size(), 0), msc(anchors.size(), 0), t(anchors.size(), 0);
    ll lb = 0;
    for (size_t i = 0; i < anchors.size(); ++i) {
        const auto a_i = anchors[i];
        const mem_t& mem_i = mems[a_i.first];
        const ll x_i = mem_i.occs[a_i.second] + mem_i.len - 1;
        const ll y_i = mem_i.rpos;
        const ll w_i = mem_i.len;
        const size_t mate_i = mem_i.mate;
        ll max_f = w_i;
        ll max_j = -1;
        size_t n_pred = 0;
        if (i - lb > (size_t)max_iter) lb = i - max_iter;
        for (ll j = i - 1; j >= lb; --j) {
            const auto a_j = anchors[j];
            const mem_t& mem_j = mems[a_j.first];
            const ll x_j = mem_j.occs[a_j.second] + mem_j.len - 1;
            const ll y_j = mem_j.rpos;
            const size_t mate_j = mem_j.mate;
            if ((mate_i != mate_j) and ((mate_i ^ mate_j) != 3)) continue;
            if (x_i > x_j + max_dist_x) { lb = j; continue; }
            const ll x_d = x_i - x_j;
            const ll y_d = y_i - y_j;
            const int32_t l = (y_d > x_d ? (y_d - x_d) : (x_d - y_d));
            const uint32_t ilog_l = (l > 0 ? ilog2_32(l) : 0);
            if ((mate_i == mate_j and (y_j >= y_i or y_d > max_dist_y)) or std::max(y_d, x_d) > G) continue;
            const ll alpha = std::min(std::min(y_d, x_d), w_i);
            ll beta = 0;
            if (mate_i != mate_j) {
                if (x_d == 0) ++beta;
                else {
                    const int c_lin = (int)(l * .01 * avg_mem_length);
                    beta = c_lin < (ll)ilog_l ? c_lin : ilog_l;       // int < uint32_t compares unsigned in the reference; both are >= 0 here
                }
            } else {
                beta = (l > 0 ? ((ll)(.01 * l * avg_mem_length) + ilog_l) >> 1 : 0);
            }
            ll score = f[j] + (alpha - beta);
            if (score > max_f) {
                max_f = score;
                max_j = j;
                if (n_pred > 0) --n_pred;
            } else if ((size_t)t[j] == i and (++n_pred > (size_t)max_pred))
                break;
            if (p[j] > 0) t[p[j]] = i;
        }
        f[i] = max_f;                                        // update_score_and_pred, chain.hpp:97-113
        p[i] = max_j;
        if (max_j >= 0 and msc[max_j] > max_f) msc[i] = msc[max_j];
        else msc[i] = max_f;
    }
    std::fill(t.begin(), t.end(), 0);
    for (size_t i = 0; i < anchors.size(); ++i) if (p[i] >= 0) t[p[i]] = 1;      // find_chain_ends
    size_t n_chains = 0;
    for (size_t i = 0; i < anchors.size(); ++i) if (t[i] == 0 and msc[i] > min_chain_score) n_chains++;
    if (n_chains == 0) return false;
    std::vector<std::pair<ll, size_t>> chain_starts(n_chains);
    size_t k = 0;
    for (size_t i = 0; i < anchors.size(); ++i) {             // find_chain_starts, chain.hpp:144-164
        if (t[i] == 0 and msc[i] > min_chain_score) {
            size_t j = i;
            while (f[j] < msc[j]) j = p[j];
            chain_starts[k++] = std::make_pair(f[j], j);
        }
    }
    chain_starts.resize(k);
    n_chains = chain_starts.size();
    std::sort(chain_starts.begin(), chain_starts.end(), std::greater<std::pair<ll, size_t>>());
    std::fill(t.begin(), t.end(), 0);
    for (size_t i = 0; i < n_chains; ++i) {                   // backtrack, chain.hpp:166-200
        ll j = chain_starts[i].second;
        chain_t chain;
        chain.mate = mems[anchors[j].first].mate;
        chain.score = chain_starts[i].first;
        do {
            chain.paired = chain.paired or (chain.mate != mems[anchors[j].first].mate);
            chain.anchors.push_back(j);
            t[j] = 1;
            j = p[j];
        } while (j >= 0 && t[j] == 0);
        if (j < 0) {
            if (chain.anchors.size() >= (size_t)min_chain_length) chains.push_back(std::move(chain));
        } else if (chain_starts[i].first - f[j] >= min_chain_score) {
            if (chain.anchors.size() >= (size_t)min_chain_length) chains.push_back(std::move(chain));
        }
    }
    auto chain_t_cmp = [](const chain_t& i, const chain_t& j) -> bool { return i.score > j.score; };
    std::sort(chains.begin(), chains.end(), chain_t_cmp);
    return true;
}

// chain.hpp:442-727 (-Z, paired-end only: aligner_ksw2.hpp:1190-1191): the same chaining with a second track per anchor - the best predecessor whose
// anchor is not on the primary chain of the current best predecessor - whose chains are added behind the primary ones before the sort by score.
// NB the chain starts of BOTH tracks are sorted by score alone here (chain.hpp:663-668), not by (score, index) as in find_chains.
inline bool find_chains_secondary(const std::vector<mem_t>& mems, std::vector<std::pair<size_t, size_t>>& anchors,
                                  std::vector<chain_t>& chains, const chain_config_t config = chain_config_t()) {
    auto cmp = [&](const std::pair<size_t, size_t>& i, const std::pair<size_t, size_t>& j) -> bool {
        return (mems[i.first].occs[i.second] + mems[i.first].len - 1) < (mems[j.first].occs[j.second] + mems[j.first].len - 1);
    };
    size_t tot_mem_length = 0;
    for (size_t i = 0; i < mems.size(); ++i) {
        for (size_t j = 0; j < mems[i].occs.size(); ++j) anchors.push_back(std::make_pair(i, j));
        tot_mem_length += mems[i].len * mems[i].occs.size();
    }
    float avg_mem_length = (float)tot_mem_length / anchors.size();
    std::sort(anchors.begin(), anchors.end(), cmp);
    const ll G = config.G;
    const ll max_dist_x = config.max_dist_x, max_dist_y = config.max_dist_y, max_iter = config.max_iter;
    const ll max_pred = config.max_pred, min_chain_score = config.min_chain_score, min_chain_length = config.min_chain_length;
    const size_t n = anchors.size();
    std::vector<ll> f(n, 0), f_sec(n, 0), p(n, 0), p_sec(n, 0), msc(n, 0), msc_sec(n, 0), t(n, 0), t_sec(n, 0);
    ll lb = 0;
    for (size_t i = 0; i < n; ++i) {
        const auto a_i = anchors[i];
        const mem_t& mem_i = mems[a_i.first];
        const ll x_i = mem_i.occs[a_i.second] + mem_i.len - 1;
        const ll y_i = mem_i.rpos;
        const ll w_i = mem_i.len;
        const size_t mate_i = mem_i.mate;
        ll max_f = w_i, max_sec_f = w_i;
        ll max_j = -1, max_sec_j = -1;
        size_t n_pred = 0;
        if (i - lb > (size_t)max_iter) lb = i - max_iter;
        for (ll j = i - 1; j >= lb; --j) {
            const auto a_j = anchors[j];
            const mem_t& mem_j = mems[a_j.first];
            const ll x_j = mem_j.occs[a_j.second] + mem_j.len - 1;
            const ll y_j = mem_j.rpos;
            const size_t mate_j = mem_j.mate;
            if ((mate_i != mate_j) and ((mate_i ^ mate_j) != 3)) continue;
            if (x_i > x_j + max_dist_x) { lb = j; continue; }
            const ll x_d = x_i - x_j;
            const ll y_d = y_i - y_j;
            const int32_t l = (y_d > x_d ? (y_d - x_d) : (x_d - y_d));
            const uint32_t ilog_l = (l > 0 ? ilog2_32(l) : 0);
            if ((mate_i == mate_j and (y_j >= y_i or y_d > max_dist_y)) or std::max(y_d, x_d) > G) continue;
            const ll alpha = std::min(std::min(y_d, x_d), w_i);
            ll beta = 0;
            if (mate_i != mate_j) {
                if (x_d == 0) ++beta;
                else {
                    const int c_lin = (int)(l * .01 * avg_mem_length);
                    beta = c_lin < (ll)ilog_l ? c_lin : ilog_l;
                }
            } else {
                beta = (l > 0 ? ((ll)(.01 * l * avg_mem_length) + ilog_l) >> 1 : 0);
            }
            ll score = f[j] + (alpha - beta);
            ll score_sec = f_sec[j] + (alpha - beta);
            if (score > max_f) {
                max_f = score;
                max_j = j;
                if (n_pred > 0) --n_pred;
            } else if (score_sec > max_sec_f) {
                if (max_j >= 0) {          // j must not lie on the primary chain that starts at max_j (chain.hpp:590-612)
                    ll tmp = max_j;
                    bool uniq_chain = true;
                    const size_t mem_j_pos = mems[anchors[j].first].occs[anchors[j].second];
                    while (tmp >= 0) {
                        const size_t mem_tmp_pos = mems[anchors[tmp].first].occs[anchors[tmp].second];
                        if (mem_j_pos == mem_tmp_pos) { uniq_chain = false; break; }
                        tmp = p[tmp];
                    }
                    if (uniq_chain) { max_sec_f = score_sec; max_sec_j = j; }
                }
            } else {
                if ((size_t)t[j] == i and (++n_pred > (size_t)max_pred)) break;
            }
            if (p[j] > 0) t[p[j]] = i;
            if (p_sec[j] > 0) t_sec[p_sec[j]] = i;
        }
        f[i] = max_f; p[i] = max_j;
        if (max_j >= 0 and msc[max_j] > max_f) msc[i] = msc[max_j]; else msc[i] = max_f;
        f_sec[i] = max_sec_f; p_sec[i] = max_sec_j;
        if (max_sec_j >= 0 and msc_sec[max_sec_j] > max_sec_f) msc_sec[i] = msc_sec[max_sec_j]; else msc_sec[i] = max_sec_f;
    }
    std::fill(t.begin(), t.end(), 0);
    std::fill(t_sec.begin(), t_sec.end(), 0);
    for (size_t i = 0; i < n; ++i) if (p[i] >= 0) t[p[i]] = 1;
    for (size_t i = 0; i < n; ++i) if (p_sec[i] >= 0) t_sec[p_sec[i]] = 1;
    size_t n_chains = 0, n_chains_sec = 0;
    for (size_t i = 0; i < n; ++i) if (t[i] == 0 and msc[i] > min_chain_score) n_chains++;
    for (size_t i = 0; i < n; ++i) if (t_sec[i] == 0 and msc_sec[i] > min_chain_score) n_chains_sec++;
    if (n_chains == 0) return false;
    auto starts_of = [&](const std::vector<ll>& tt, const std::vector<ll>& ff, const std::vector<ll>& pp, const std::vector<ll>& mm, size_t cnt) {
        std::vector<std::pair<ll, size_t>> cs(cnt);
        size_t k = 0;
        for (size_t i = 0; i < n; ++i) {
            if (tt[i] == 0 and mm[i] > min_chain_score) {
                size_t j = i;
                while (ff[j] < mm[j]) j = pp[j];
                cs[k++] = std::make_pair(ff[j], j);
            }
        }
        cs.resize(k);
        return cs;
    };
    auto chain_starts = starts_of(t, f, p, msc, n_chains);
    auto chain_starts_sec = starts_of(t_sec, f_sec, p_sec, msc_sec, n_chains_sec);
    auto chain_start_cmp = [](const std::pair<ll, size_t>& lhs, const std::pair<ll, size_t>& rhs) { return lhs.first > rhs.first; };
    std::sort(chain_starts.begin(), chain_starts.end(), chain_start_cmp);
    std::sort(chain_starts_sec.begin(), chain_starts_sec.end(), chain_start_cmp);
    auto backtrack = [&](const std::vector<std::pair<ll, size_t>>& cs, std::vector<ll>& tt, const std::vector<ll>& ff, const std::vector<ll>& pp) {
        std::fill(tt.begin(), tt.end(), 0);
        for (size_t i = 0; i < cs.size(); ++i) {
            ll j = cs[i].second;
            chain_t chain;
            chain.mate = mems[anchors[j].first].mate;
            chain.score = cs[i].first;
            do {
                chain.paired = chain.paired or (chain.mate != mems[anchors[j].first].mate);
                chain.anchors.push_back(j);
                tt[j] = 1;
                j = pp[j];
            } while (j >= 0 && tt[j] == 0);
            if (j < 0) {
                if (chain.anchors.size() >= (size_t)min_chain_length) chains.push_back(std::move(chain));
            } else if (cs[i].first - ff[j] >= min_chain_score) {
                if (chain.anchors.size() >= (size_t)min_chain_length) chains.push_back(std::move(chain));
            }
        }
    };
    backtrack(chain_starts, t, f, p);
    backtrack(chain_starts_sec, t_sec, f_sec, p_sec);
    auto chain_t_cmp = [](const chain_t& i, const chain_t& j) -> bool { return i.score > j.score; };
    std::sort(chains.begin(), chains.end(), chain_t_cmp);
    return true;
}

// ---- mapq.hpp:146-184 -------------------------------------------------------------------------------------
inline size_t compute_mapq_se_bwa(const int32_t score, const int32_t score2, const int32_t rlen, const int32_t qlen,
                                  const int32_t min_seed_length, const int32_t match_score, const int32_t mismatch_score,
                                  const double mapq_coeff_len, const int32_t mapq_coeff_fac, const int32_t sub_n,
                                  const int32_t seed_cov, const double frac_rep) {
    int32_t mapq = 0;
    int32_t l = std::max(rlen, qlen);
    int32_t sub = score2 ? score2 : min_seed_length * match_score;
    if (sub >= score) return mapq;
    double identity = 1. - (double)(l * match_score - score) / (match_score + mismatch_score) / l;
    if (score == 0) {
        mapq = 0;
    } else if (mapq_coeff_len > 0) {
        double tmp;
        tmp = l < mapq_coeff_len ? 1. : mapq_coeff_fac / log(l);
        tmp *= identity * identity;
        mapq = (int)(6.02 * (score - sub) / match_score * tmp * tmp + .499);
    } else {
        mapq = (int)(30.0 * (1. - (double)sub / score) * log(seed_cov) + .499);
        mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
    }
    if (sub_n > 0) mapq -= (int)(4.343 * log(sub_n + 1) + .499);
    if (mapq > 60) mapq = 60;
    if (mapq < 0) mapq = 0;
    mapq = (int)(mapq * (1. - frac_rep) + .499);
    return mapq;
}

// ---- sam.hpp ------------------------------------------------------------------------------------------------
struct read_t {                    // the kseq_t fields the path touches
    std::string name, seq, qual;
    bool has_qual = false;
};

struct sam_t {                     // sam.hpp:47-112
    const read_t* read = nullptr;
    size_t flag = 4, pos = 0, mapq = 255, pnext = 0;
    long long int tlen = 0;
    std::string rname = "*", cigar = "*", rnext = "*";
    size_t as = 0, nm = 0, zs = 0;
    std::string md = "", oa = "", aa = "";
    std::vector<std::string> alt_haplotypes;
    std::vector<size_t> alt_pos, alt_scores;
    size_t rlen = 0;
    std::string lift_rname = "*", lift_cigar = "*";
    size_t lift_pos = 0, lift_mapq = 0, lift_nm = 0;
    std::string lift_md = "";
    size_t lift_rlen = 0;
    bool unmapped_lft = false;
};

// sam.hpp:144-188.  The reference prints size_t values with "%d": the low 32 bits as a signed int.
inline void write_sam(std::string& out, const sam_t& s) {
    char buf[64];
    auto d = [&](size_t v) { snprintf(buf, sizeof buf, "%d", (int)v); out += buf; };
    out += s.read->name; out += '\t';
    d(s.flag); out += '\t';
    out += s.rname; out += '\t';
    d(s.pos); out += '\t';
    d(s.mapq); out += '\t';
    out += s.cigar; out += '\t';
    out += s.rnext; out += '\t';
    d(s.pnext); out += '\t';
    d((size_t)s.tlen); out += '\t';
    out += s.read->seq; out += '\t';
    if (s.read->has_qual) out += s.read->qual; else out += "*";
    if (!(s.flag & 4) or s.unmapped_lft) {
        out += "\tAS:i:"; d(s.as);
        out += "\tNM:i:"; d(s.nm);
        if (s.zs > 0) { out += "\tZS:i:"; d(s.zs); }
        out += "\tMD:Z:"; out += s.md;
        out += "\tOA:Z:"; out += s.lift_rname; out += ',';
        d(s.lift_pos); out += ',';
        out += (s.flag & 16) ? "-," : "+,";
        out += s.lift_cigar; out += ',';
        d(s.mapq); out += ',';
        d(s.lift_nm); out += ';';
        out += "\tAA:Z:";
        for (size_t i = 0; i < s.alt_haplotypes.size(); i++) {
            out += s.alt_haplotypes[i]; out += ','; d(s.alt_pos[i]); out += ','; d(s.alt_scores[i]); out += ';';
        }
    }
    out += '\n';
}

// sam.hpp:249-287
inline size_t write_MD_core(const uint8_t* tseq, const uint8_t* qseq, const uint32_t* cigar, const size_t n_cigar, std::string& mdz) {
    int i, q_off, t_off, l_MD = 0, NM = 0;
    for (i = q_off = t_off = 0; i < (int)n_cigar; ++i) {
        int j, op = cigar[i] & 0xf, len = cigar[i] >> 4;
        if (op == 0 || op == 7 || op == 8) {
            for (j = 0; j < len; ++j) {
                if (qseq[q_off + j] != tseq[t_off + j]) {
                    mdz += std::to_string(l_MD) + "ACGTN"[tseq[t_off + j]];
                    l_MD = 0;
                    ++NM;
                } else ++l_MD;
            }
            q_off += len, t_off += len;
        } else if (op == 1) {
            q_off += len;
            NM += len;
        } else if (op == 2) {
            mdz += std::to_string(l_MD) + "^";
            for (j = 0; j < len; ++j) mdz.push_back("ACGTN"[tseq[t_off + j]]);
            l_MD = 0;
            t_off += len;
            NM += len;
        } else if (op == 3) {
            t_off += len;
        }
    }
    if (l_MD > 0) mdz += std::to_string(l_MD);
    return NM;
}

// ---- aligner ----------------------------------------------------------------------------------------------------
struct align_config_t {            // aligner_ksw2.hpp:84-130 with the `moni align` wrapper defaults (pipeline/moni.in:748-768)
    size_t min_len = 25, ext_len = 100, check_k = 5, region_dist = 10;
    bool filter_seeds = true;
    size_t n_seeds_thr = 1000;
    bool filter_freq = true;
    double freq_thr = 0.50;
    int8_t smatch = 2, smismatch = 4, gapo = 4, gapo2 = 13, gape = 2, gape2 = 1;
    int end_bonus = 400, w = -1, zdrop = -1;
    bool report_mems = false, left_mem_check = true;
    chain_config_t chain;
};

struct score_t {                   // aligner_ksw2.hpp:134-139
    int32_t score = 0;
    size_t pos = 0, lft = 0;
    bool unmapped_lft = false;
};

struct dp_counters { uint64_t cells = 0, calls = 0, ref_bytes = 0; };

struct aligner {
    const FlatIndex& ix;
    align_config_t cfg;
    seed_finder mem_finder;
    size_t n;
    const int m = 5;
    int8_t mat[25];
    float mapq_coeff_len = 50.0;
    int32_t mapq_coeff_fac = log(50.0);        // aligner_ksw2.hpp:3250-3251: (int32_t)log(50.0f) = 3
    unsigned char seq_nt4_table[256];
    dp_counters dpc;
    ksw_counters kc;

    aligner(const FlatIndex& ix_, const align_config_t& c) : ix(ix_), cfg(c), mem_finder(ix_, c.min_len, c.filter_seeds, c.n_seeds_thr), n(ix_.n_text) {
        // ksw_gen_simple_mat, aligner_ksw2.hpp:3199-3211
        int i, j;
        int8_t a = cfg.smatch, b = -cfg.smismatch;
        a = a < 0 ? -a : a; b = b > 0 ? -b : b;
        for (i = 0; i < m - 1; ++i) {
            for (j = 0; j < m - 1; ++j) mat[i * m + j] = i == j ? a : b;
            mat[i * m + m - 1] = 0;
        }
        for (j = 0; j < m; ++j) mat[(m - 1) * m + j] = 0;
        for (i = 0; i < 256; ++i) seq_nt4_table[i] = 4;
        seq_nt4_table[0] = 0; seq_nt4_table[1] = 1; seq_nt4_table[2] = 2; seq_nt4_table[3] = 3;   // aligner_ksw2.hpp:3273
        seq_nt4_table['A'] = seq_nt4_table['a'] = 0; seq_nt4_table['C'] = seq_nt4_table['c'] = 1;
        seq_nt4_table['G'] = seq_nt4_table['g'] = 2; seq_nt4_table['T'] = seq_nt4_table['t'] = 3;
    }

    // ra.expandSubstr + nt4 conversion; positions past the text end read as separators (never reached in practice)
    void expand_nt4(size_t pos, size_t len, uint8_t* dst, bool reversed = false) {
        dpc.ref_bytes += len;
        for (size_t i = 0; i < len; ++i) {
            uint8_t ch = (pos + i) < ix.text.size() ? ix.text[pos + i] : 0;
            if (reversed) dst[len - i - 1] = seq_nt4_table[ch];
            else dst[i] = seq_nt4_table[ch];
        }
    }

    void extz(int qlen, const uint8_t* q, int tlen, const uint8_t* t, int flag, ksw_extz_t* ez) {
        ksw_extz2_restated(qlen, q, tlen, t, m, mat, cfg.gapo, cfg.gape, cfg.w, cfg.zdrop, cfg.end_bonus, flag, ez, &kc);
    }

    // aligner_ksw2.hpp:2752-3196
    score_t fill_chain(const std::vector<mem_t>& mems, const std::vector<std::pair<size_t, size_t>>& anchors, const uint8_t* lcs,
                       const size_t lcs_len, const uint8_t* rcs, const size_t rcs_len, const read_t* read, const bool score_only = true,
                       sam_t* sam = nullptr, bool realign = false) {
        const size_t ext_len = cfg.ext_len;
        const int8_t smatch = cfg.smatch, gapo = cfg.gapo, gape = cfg.gape, gapo2 = cfg.gapo2, gape2 = cfg.gape2;
        int flag = KSW_EZ_EXTZ_ONLY | KSW_EZ_RIGHT;
        if (score_only) flag = KSW_EZ_SCORE_ONLY;
        score_t score;
        int score_lc = 0, score_rc = 0;
        ksw_extz_t ez_lc, ez_rc, ez;
        memset(&ez_lc, 0, sizeof(ksw_extz_t)); memset(&ez_rc, 0, sizeof(ksw_extz_t)); memset(&ez, 0, sizeof(ksw_extz_t));
        ksw_reset_extz(&ez_lc);
        if (lcs_len > 0) {
            size_t mem_pos = mems[anchors[0].first].occs[anchors[0].second];
            size_t lc_occ = (mem_pos > ext_len ? mem_pos - ext_len : 0);
            size_t lc_len = (mem_pos > ext_len ? ext_len : ext_len - mem_pos);
            std::vector<uint8_t> lc(ext_len + 1);
            expand_nt4(lc_occ, lc_len, lc.data(), true);
            extz(lcs_len, lcs, lc_len, lc.data(), flag, &ez_lc);
            score_lc = ez_lc.mqe;
        }
        ksw_reset_extz(&ez_rc);
        if (rcs_len > 0) {
            size_t mem_pos = mems[anchors.back().first].occs[anchors.back().second];
            size_t mem_len = mems[anchors.back().first].len;
            size_t rc_occ = mem_pos + mem_len;
            size_t rc_len = (rc_occ < n - ext_len ? ext_len : n - rc_occ);
            std::vector<uint8_t> rc(std::max(ext_len, rc_len) + 1);
            expand_nt4(rc_occ, rc_len, rc.data());
            extz(rcs_len, rcs, rc_len, rc.data(), flag, &ez_rc);
            score_rc = ez_rc.mqe;
        }
        score.score = score_lc + score_rc;
        size_t mem_pos = mems[anchors[0].first].occs[anchors[0].second];
        size_t mem_len = mems[anchors.back().first].occs[anchors.back().second] + mems[anchors.back().first].len - mem_pos;
        size_t ref_pos;
        if ((size_t)(lcs_len > 0 ? ez_lc.mqe_t + 1 : 0) > mem_pos) ref_pos = 0;
        else ref_pos = mem_pos - (lcs_len > 0 ? ez_lc.mqe_t + 1 : 0);
        size_t ref_len = (lcs_len > 0 ? ez_lc.mqe_t + 1 : 0) + mem_len + (rcs_len > 0 ? ez_rc.mqe_t + 1 : 0);
        std::vector<uint8_t> refv(ref_len + 1);
        uint8_t* ref = refv.data();
        expand_nt4(ref_pos, ref_len, ref);
        size_t seq_len = read->seq.size();
        std::vector<uint8_t> seqv(seq_len + 1);
        uint8_t* seq = seqv.data();
        for (size_t i = 0; i < seq_len; ++i) seq[i] = seq_nt4_table[(unsigned char)read->seq[i]];
        score.pos = ref_pos;

        bool mems_overlap = false;
        size_t last_ref = mem_pos + mems[anchors[0].first].len;
        size_t last_seq = mems[anchors[0].first].idx + mems[anchors[0].first].len;
        for (size_t i = 1; i < anchors.size() and not mems_overlap; ++i) {
            const size_t& ref_occ = mems[anchors[i].first].occs[anchors[i].second];
            const size_t& seq_occ = mems[anchors[i].first].idx;
            const size_t& mlen = mems[anchors[i].first].len;
            if (last_ref > ref_occ or last_seq > seq_occ) mems_overlap = true;
            last_ref = ref_occ + mlen;
            last_seq = seq_occ + mlen;
        }
        std::vector<ksw_extz_t> ez_cc(anchors.size() - 1);
        for (auto& x : ez_cc) memset(&x, 0, sizeof(ksw_extz_t));
        if (not mems_overlap and not realign) {
            size_t last_ref = mem_pos + mems[anchors[0].first].len;
            size_t last_seq = mems[anchors[0].first].idx + mems[anchors[0].first].len;
            for (size_t i = 1; i < anchors.size(); ++i) {
                const size_t& ref_occ = mems[anchors[i].first].occs[anchors[i].second];
                const size_t& seq_occ = mems[anchors[i].first].idx;
                const size_t& mlen = mems[anchors[i].first].len;
                if (last_ref == ref_occ) {
                    if (last_seq < seq_occ) {
                        size_t l = (seq_occ - last_seq);
                        ez_cc[i - 1].score = -std::min(gapo + l * gape, gapo2 + l * gape2);
                        ez_cc[i - 1].m_cigar = 1; ez_cc[i - 1].n_cigar = 1;
                        ez_cc[i - 1].cigar = (uint32_t*)malloc(sizeof(uint32_t));
                        ez_cc[i - 1].cigar[0] = (l << 4) | 1;
                    } else {
                        ez_cc[i - 1].score = 0; ez_cc[i - 1].m_cigar = 0; ez_cc[i - 1].n_cigar = 0;
                    }
                } else {
                    if (last_seq == seq_occ) {
                        size_t l = (seq_occ - last_seq);       // = 0 in the reference (aligner_ksw2.hpp:2939); reproduced
                        ez_cc[i - 1].score = -std::min(gapo + l * gape, gapo2 + l * gape2);
                        ez_cc[i - 1].m_cigar = 1; ez_cc[i - 1].n_cigar = 1;
                        ez_cc[i - 1].cigar = (uint32_t*)malloc(sizeof(uint32_t));
                        ez_cc[i - 1].cigar[0] = (l << 4) | 2;
                    } else {
                        flag = KSW_EZ_RIGHT;                   // sticks for the rest of this call, also in score-only mode
                        ksw_reset_extz(&ez_cc[i - 1]);
                        size_t cc_occ = mems[anchors[i - 1].first].occs[anchors[i - 1].second] + mems[anchors[i - 1].first].len;
                        size_t cc_len = mems[anchors[i].first].occs[anchors[i].second] - cc_occ;
                        cc_occ -= ref_pos;
                        size_t ccs_pos = mems[anchors[i - 1].first].idx + mems[anchors[i - 1].first].len;
                        size_t ccs_len = mems[anchors[i].first].idx - ccs_pos;
                        extz(ccs_len, seq + ccs_pos, cc_len, ref + cc_occ, flag, &ez_cc[i - 1]);
                    }
                }
                last_ref = ref_occ + mlen;
                last_seq = seq_occ + mlen;
                score.score += mems[anchors[i - 1].first].len * smatch + ez_cc[i - 1].score;
            }
            score.score += mems[anchors.back().first].len * smatch;
        } else {
            ksw_reset_extz(&ez);
            extz(seq_len, seq, ref_len, ref, flag, &ez);
            score.score = ez.score;
            realign = true;
        }
        bool is_valid = ix.valid(ref_pos, ref_len);
        if (not is_valid) score.score = std::numeric_limits<int32_t>::min();
        if (is_valid and not score_only) {
            size_t n_cigar = 0;
            uint32_t* cigar = nullptr;
            if (realign) {
                flag = KSW_EZ_RIGHT;
                free(ez.cigar); ez.cigar = nullptr; ez.m_cigar = 0;      // (the reference leaks the first CIGAR here)
                ksw_reset_extz(&ez);
                extz(seq_len, seq, ref_len, ref, flag, &ez);
                n_cigar = ez.n_cigar;
                cigar = ez.cigar;
                score.score = ez.score;
                ez.m_cigar = 0; ez.cigar = nullptr;
            } else {
                n_cigar = ez_lc.n_cigar + ez_rc.n_cigar + 1;
                for (size_t i = 0; i < anchors.size() - 1; ++i) n_cigar += ez_cc[i].n_cigar + 1;
                cigar = (uint32_t*)calloc(n_cigar, sizeof(uint32_t));
                size_t i = 0;
                for (size_t j = 0; j < (size_t)ez_lc.n_cigar; ++j) cigar[i++] = ez_lc.cigar[ez_lc.n_cigar - j - 1];
                for (size_t j = 0; j < anchors.size(); ++j) {
                    const size_t& mlen = mems[anchors[j].first].len;
                    if (i > 0 and ((cigar[i - 1] & 0xf) == 0)) { cigar[i - 1] += (((uint32_t)mlen) << 4); --n_cigar; }
                    else cigar[i++] = (((uint32_t)mlen) << 4);
                    if (j < anchors.size() - 1) {
                        if (ez_cc[j].n_cigar > 0) {
                            if ((ez_cc[j].cigar[0] & 0xf) == 0) { cigar[i - 1] += ez_cc[j].cigar[0]; --n_cigar; }
                            else cigar[i++] = ez_cc[j].cigar[0];
                        }
                        for (size_t k = 1; k < (size_t)ez_cc[j].n_cigar; ++k) cigar[i++] = ez_cc[j].cigar[k];
                    }
                }
                if (ez_rc.n_cigar > 0) {
                    if ((ez_rc.cigar[0] & 0xf) == 0) { cigar[i - 1] += ez_rc.cigar[0]; --n_cigar; }
                    else cigar[i++] = ez_rc.cigar[0];
                }
                for (size_t j = 1; j < (size_t)ez_rc.n_cigar; ++j) cigar[i++] = ez_rc.cigar[j];
            }
            sam->lift_cigar = "";
            for (size_t i = 0; i < n_cigar; ++i) sam->lift_cigar += std::to_string(cigar[i] >> 4) + "MID"[cigar[i] & 0xf];
            sam->lift_nm = write_MD_core(ref, seq, cigar, n_cigar, sam->lift_md);
            const auto refi = ix.index(ref_pos);
            sam->as = score.score;
            sam->lift_pos = refi.second + 1;
            sam->lift_rname = ix.names[refi.first];
            sam->lift_rlen = ref_len;
            // bam_set1(.., pos = ref.second, .., n_cigar, cigar, ..) + idx.lift_cigar(bam, ref_pos)
            const std::vector<uint32_t> lft_cigar = ix.lift_cigar(cigar, n_cigar, ref_pos);
            const auto lift = ix.lift(ref_pos);
            const auto lft_ref = ix.index(lift);
            sam->pos = lft_ref.second + 1;
            sam->rname = ix.names[lft_ref.first];
            sam->cigar = "";
            for (size_t i = 0; i < lft_cigar.size(); ++i) sam->cigar += std::to_string(lft_cigar[i] >> 4) + "MID"[lft_cigar[i] & 0xf];
            ref_pos = lift;
            ref_len = 0;                                   // bam_cigar2rlen: M, D, N, =, X consume the reference
            for (size_t i = 0; i < lft_cigar.size(); ++i) { int op = lft_cigar[i] & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += lft_cigar[i] >> 4; }
            if (ref_len > 0) {
                std::vector<uint8_t> l_ref(ref_len + 1);
                expand_nt4(ref_pos, ref_len, l_ref.data());
                sam->nm = write_MD_core(l_ref.data(), seq, lft_cigar.data(), lft_cigar.size(), sam->md);
                sam->rlen = ref_len;
                score.score = ez.score;                    // as in the reference: ez is the (possibly never used) global result
                score.pos = ref_pos;
            } else {
                sam->pos = 0; sam->rname = "*"; sam->cigar = "*"; sam->rlen = 0;
                sam->unmapped_lft = true;
                score.unmapped_lft = true;
            }
            free(cigar);
        }
        if (ez_lc.m_cigar > 0) free(ez_lc.cigar);
        if (ez_rc.m_cigar > 0) free(ez_rc.cigar);
        if (ez.m_cigar > 0) free(ez.cigar);
        for (size_t i = 0; i < ez_cc.size(); ++i) if (ez_cc[i].n_cigar > 0) free(ez_cc[i].cigar);
        return score;
    }

    // aligner_ksw2.hpp:2018-2098
    score_t chain_score(const std::vector<size_t>& chain, const std::vector<std::pair<size_t, size_t>>& anchors,
                        const std::vector<mem_t>& mems, const int32_t min_score, const read_t* read, const bool score_only = true,
                        const int32_t score2 = 0, const uint8_t strand = 0, sam_t* sam = nullptr, const int32_t sub_n = 0, const double frac_rep = 0) {
        std::vector<std::pair<size_t, size_t>> chain_anchors(chain.size());
        for (size_t i = 0; i < chain_anchors.size(); ++i) chain_anchors[i] = anchors[chain[i]];
        size_t lcs_len = mems[chain_anchors[0].first].idx;
        std::vector<uint8_t> lcs(lcs_len + 1);
        for (size_t i = 0; i < lcs_len; ++i) lcs[lcs_len - i - 1] = seq_nt4_table[(unsigned char)read->seq[i]];
        size_t rcs_occ = (mems[chain_anchors.back().first].idx + mems[chain_anchors.back().first].len);
        size_t rcs_len = read->seq.size() - rcs_occ;
        std::vector<uint8_t> rcs(rcs_len + 1);
        for (size_t i = 0; i < rcs_len; ++i) rcs[i] = seq_nt4_table[(unsigned char)read->seq[rcs_occ + i]];
        score_t score = fill_chain(mems, chain_anchors, lcs.data(), lcs_len, rcs.data(), rcs_len, read);
        if (!score_only and score.score >= min_score) {
            auto tmp_score = fill_chain(mems, chain_anchors, lcs.data(), lcs_len, rcs.data(), rcs_len, read, score_only, sam);
            score.unmapped_lft = tmp_score.unmapped_lft;
            sam->flag = (strand ? 16 : 0);
            sam->zs = score2;
            sam->mapq = compute_mapq_se_bwa(sam->as, sam->zs, sam->rlen, read->seq.size(), cfg.min_len, cfg.smatch, cfg.smismatch,
                                            mapq_coeff_len, mapq_coeff_fac, sub_n, 0, frac_rep);
        }
        return score;
    }

    // include/common/csv.hpp:26-52
    struct csv_t {
        size_t num_uniq_mems = 0, total_mem_occ = 0;
        double max_mem_freq = 0, min_mem_freq = 1;
        size_t high_occ_mem = 0, low_occ_mem = 0, num_mems_filter = 0, num_chains_skipped = 0;
    };
    // aligner_ksw2.hpp:1868-1902
    void calculate_MEM_stats(const std::vector<mem_t>& mems, csv_t& csv) {
        csv.num_uniq_mems = mems.size();
        for (size_t i = 0; i < mems.size(); ++i) {
            csv.total_mem_occ += mems[i].total_occ;
            csv.num_mems_filter += mems[i].num_filtered;
        }
        for (size_t i = 0; i < mems.size(); ++i) {
            double mem_freq = (mems[i].occs.size() / (static_cast<double>(csv.total_mem_occ)));
            csv.max_mem_freq = (csv.max_mem_freq > mem_freq ? csv.max_mem_freq : mem_freq);
            csv.min_mem_freq = (csv.min_mem_freq > mem_freq ? mem_freq : csv.min_mem_freq);
            for (auto it = mems[i].count_dict.begin(); it != mems[i].count_dict.end(); ++it) {
                if (csv.high_occ_mem == 0 && csv.low_occ_mem == 0) { csv.high_occ_mem = it->second; csv.low_occ_mem = it->second; }
                else {
                    csv.high_occ_mem = (csv.high_occ_mem > it->second ? csv.high_occ_mem : it->second);
                    csv.low_occ_mem = (csv.low_occ_mem > it->second ? it->second : csv.low_occ_mem);
                }
            }
        }
    }
    // include/common/csv.hpp:55-67
    static void write_csv(std::string& out, const std::string& name, const csv_t& c) {
        char buf[256];
        out += name;
        snprintf(buf, sizeof buf, ",%zu,%zu,%f,%f,%zu,%zu,%zu,%zu\n", c.num_uniq_mems, c.total_mem_occ, c.max_mem_freq, c.min_mem_freq, c.high_occ_mem, c.low_occ_mem,
                 c.num_mems_filter, c.num_chains_skipped);
        out += buf;
    }

    // aligner_ksw2.hpp:1905-1933
    void seed_freq_filter(std::vector<mem_t>& mems, const double freq, csv_t& csv) {
        size_t total_mem_occ = 0;
        std::vector<size_t> delete_ind;
        for (size_t i = 0; i < mems.size(); ++i) total_mem_occ += mems[i].occs.size();
        for (size_t i = 0; i < mems.size(); ++i) {
            double mem_freq = (static_cast<double>(mems[i].occs.size()) / total_mem_occ);
            if (mem_freq > freq) { delete_ind.push_back(i); csv.num_mems_filter += (mems[i].occs.size()); }
        }
        std::reverse(delete_ind.begin(), delete_ind.end());
        for (size_t idx : delete_ind) mems.erase(mems.begin() + idx);
    }

    struct alignment_t {
        bool aligned = false, chained = false, best_score = false;
        const read_t* read;
        read_t read_rev;
        sam_t sam;
        score_t score;
        int32_t score2 = 0;
        int sub_n = 0;
        std::vector<mem_t> mems;
        std::vector<std::pair<size_t, size_t>> anchors;
        std::vector<chain_t> chains;
        std::string mems_sam;          // report_mems: the records of aligner_ksw2.hpp:346-373
        csv_t csv;                     // -c: the MEM statistics (aligner_ksw2.hpp:154, 340-343, 417)
    };

    // aligner_ksw2.hpp:553-597
    bool check_left_MEM(std::vector<std::pair<size_t, size_t>>& left_mem_vec, alignment_t& al, size_t i) {
        auto& chain = al.chains[i];
        chain.reverse();
        size_t left_mem_pos, left_mem_ref_pos = 0;
        for (size_t j = 0; j < chain.anchors.size(); ++j) {
            size_t anchor_id = chain.anchors[j];
            left_mem_pos = al.mems[al.anchors[anchor_id].first].occs[al.anchors[anchor_id].second];
            const auto lift = ix.lift(left_mem_pos);
            const auto lft_ref = ix.index(lift);
            left_mem_ref_pos = lft_ref.second + 1;
            break;
        }
        bool discovered = false;
        for (size_t j = 0; j < left_mem_vec.size(); ++j) {
            if (ORC_DIST(left_mem_vec[j].first, left_mem_ref_pos) < cfg.region_dist) {
                if (left_mem_vec[j].second == (size_t)al.chains[i].score) discovered = true;
            }
        }
        chain.reset();
        if (discovered) return true;
        left_mem_vec.push_back(std::make_pair(left_mem_ref_pos, (size_t)al.chains[i].score));
        return false;
    }

    // aligner_ksw2.hpp:528-548
    int32_t check_max_score(int32_t max_score, const score_t& s, std::vector<std::string>& alt_haplotypes,
                            std::vector<size_t>& alt_pos, std::vector<size_t>& alt_scores) {
        if (s.score > max_score) {
            max_score = s.score;
            alt_haplotypes.clear(); alt_pos.clear(); alt_scores.clear();
        } else if (s.score == max_score) {
            auto ref = ix.index(s.pos);
            alt_haplotypes.emplace_back(ix.names[ref.first]);
            alt_pos.emplace_back(ref.second + 1);
            alt_scores.emplace_back(s.score);
        }
        return max_score;
    }

    // aligner_ksw2.hpp:328-521
    bool align(alignment_t& al) {
        mem_finder.find_mems(al.read->seq.data(), al.read->seq.size(), al.mems, 0, MATE_1 | MATE_F);
        mem_finder.find_mems(al.read_rev.seq.data(), al.read_rev.seq.size(), al.mems, 0, MATE_1 | MATE_RC);
        mem_finder.populate_seeds(al.mems, cfg.report_mems);
        calculate_MEM_stats(al.mems, al.csv);                                  // (if (csv): the statistics change nothing else)
        if (cfg.filter_freq) seed_freq_filter(al.mems, cfg.freq_thr, al.csv);
        if (cfg.report_mems) {                      // aligner_ksw2.hpp:346-373: one secondary record per occurrence of every MEM
            for (size_t i = 0; i < al.mems.size(); ++i) {
                const read_t& src = (al.mems[i].mate & MATE_RC) ? al.read_rev : *al.read;
                read_t part;                        // copy_partial_kseq_t (kpbseq.h:197-205)
                part.name = src.name; part.has_qual = src.has_qual;
                part.seq = src.seq.substr(al.mems[i].idx, al.mems[i].len);
                if (src.has_qual) part.qual = src.qual.substr(al.mems[i].idx, al.mems[i].len);
                for (size_t j = 0; j < al.mems[i].occs.size(); ++j) {
                    sam_t rs;
                    rs.read = &part;
                    rs.cigar = std::to_string(al.mems[i].len) + "M";
                    const auto ref = ix.index(al.mems[i].occs[j]);
                    rs.pos = ref.second + 1;
                    rs.rname = ix.names[ref.first];
                    rs.flag = (al.mems[i].mate & MATE_RC) ? (256 | 16) : 256;
                    write_sam(al.mems_sam, rs);
                }
            }
            al.aligned = true;
            return true;
        }
        al.chained = find_chains(al.mems, al.anchors, al.chains, cfg.chain);
        if (not al.chained) return false;
        int32_t min_score = 20 + 8 * log(al.read->seq.size());
        std::vector<std::tuple<int32_t, size_t, size_t>> best_scores;
        std::set<size_t> different_scores;
        size_t i = 0;
        std::vector<std::pair<size_t, size_t>> left_mem_vec;
        int32_t max_score = 0;
        std::vector<std::string> alt_haplotypes;
        std::vector<size_t> alt_pos, alt_scores;
        while (i < al.chains.size() and different_scores.size() < cfg.check_k) {
            different_scores.insert(al.chains[i].score);
            if (cfg.left_mem_check) {
                if (check_left_MEM(left_mem_vec, al, i)) { ++i; al.csv.num_chains_skipped++; continue; }
            }
            if (different_scores.size() < cfg.check_k) {
                auto chain = al.chains[i];
                std::reverse(chain.anchors.begin(), chain.anchors.end());
                score_t score;
                if ((chain.mate & MATE_RC)) score = chain_score(chain.anchors, al.anchors, al.mems, min_score, &al.read_rev);
                else score = chain_score(chain.anchors, al.anchors, al.mems, min_score, al.read);
                score.lft = ix.lift(score.pos);
                max_score = check_max_score(max_score, score, alt_haplotypes, alt_pos, alt_scores);
                bool replaced = false;
                for (size_t j = 0; j < best_scores.size(); ++j) {
                    if ((ORC_DIST(std::get<1>(best_scores[j]), score.lft) < cfg.region_dist)) {
                        if (score.score > std::get<0>(best_scores[j])) {
                            if (replaced) best_scores[j] = std::make_tuple(0, 0, i - 1);
                            else { best_scores[j] = std::make_tuple(score.score, score.lft, i); i++; replaced = true; }
                        } else if (score.score <= std::get<0>(best_scores[j])) {
                            j = best_scores.size(); replaced = true; i++;
                        }
                    }
                }
                if (not replaced) { best_scores.push_back(std::make_tuple(score.score, score.lft, i)); i++; }
            }
        }
        al.sam.alt_haplotypes = alt_haplotypes;
        al.sam.alt_pos = alt_pos;
        al.sam.alt_scores = alt_scores;
        al.sub_n = best_scores.size() - 1;
        while (best_scores.size() < 2) best_scores.push_back(std::make_tuple(0, 0, al.chains.size()));
        std::sort(best_scores.begin(), best_scores.end(), std::greater<std::tuple<int32_t, size_t, size_t>>());
        if (std::get<0>(best_scores[0]) < min_score) return false;
        al.best_score = true;
        al.score2 = std::get<0>(best_scores[1]);
        {
            i = std::get<2>(best_scores[0]);
            auto chain = al.chains[i];
            std::reverse(chain.anchors.begin(), chain.anchors.end());
            if ((chain.mate & MATE_RC)) {
                al.score = chain_score(chain.anchors, al.anchors, al.mems, min_score, &al.read_rev, false, al.score2, 1, &al.sam);
                al.sam.read = &al.read_rev;
                al.sam.flag |= 16;
            } else
                al.score = chain_score(chain.anchors, al.anchors, al.mems, min_score, al.read, false, al.score2, 0, &al.sam);
        }
        al.aligned = (al.score.score >= min_score);
        return al.aligned;
    }

    // aligner_ksw2.hpp:314-325 + alignment_t ctor 169-176: one read -> one SAM line
    bool align_read(const read_t& read, std::string& out, std::string* csv_out = nullptr) {
        static const unsigned char* ct = nullptr;
        static unsigned char ctab[256];
        if (!ct) {
            for (int i = 0; i < 256; ++i) ctab[i] = (unsigned char)i;
            ctab['A'] = 'T'; ctab['C'] = 'G'; ctab['G'] = 'C'; ctab['T'] = 'A';
            ctab['a'] = 'T'; ctab['c'] = 'G'; ctab['g'] = 'C'; ctab['t'] = 'A';
            ct = ctab;
        }
        alignment_t al;
        al.read = &read;
        al.sam.read = &read;
        al.read_rev.name = read.name;
        al.read_rev.has_qual = read.has_qual;
        const size_t l = read.seq.size();
        al.read_rev.seq.resize(l);
        for (size_t i = 0; i < l; ++i) al.read_rev.seq[i] = (char)ct[(unsigned char)read.seq[l - i - 1]];
        al.read_rev.qual.assign(read.qual.rbegin(), read.qual.rend());
        if (not align(al)) al.sam.flag = 4;               // set_sam_not_aligned
        if (!cfg.report_mems) write_sam(out, al.sam);
        else out += al.mems_sam;
        if (csv_out) write_csv(*csv_out, read.name, al.csv);                  // alignment.record_csv (aligner_ksw2.hpp:322-323)
        return al.aligned;
    }

    // aligner_ksw2.hpp:3213-3219 + seqidx.hpp:174-180
    std::string sam_header() const {
        std::string res = "@HD\tVN:1.6\tSO:unknown\n";
        for (size_t i = 0; i < ix.names.size(); ++i) res += "@SQ\tSN:" + ix.names[i] + "\tLN:" + std::to_string(ix.seq_length(i)) + "\n";
        res += "@PG\tID:moni\tPN:moni\tVN:0.1.0\n";
        return res;
    }
};

}  // namespace oracle
