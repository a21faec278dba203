// Host part of the paired-end path: what follows the kernel's decision for a pair (pe_core.h) - lift-over of the two CIGARs, MD/NM,
// the single-end MAPQ of each mate, then the tail of paired_chain_score (aligner_ksw2.hpp:2200-2288: PNEXT / TLEN / flags / the
// paired MAPQ, or one mate placed by the other) and the two SAM lines (sam.hpp:144-188).
#pragma once
#include <cmath>
#include <string>

#include "align_host.hpp"

namespace mh {

struct PeMateOut {
    uint32_t m = 0; uint64_t off = 0;            // the mate in the batch
    int32_t score = 0;                            // score.m1 / score.m2 of the best pair (score-only pass)
    bool filled = false;                          // chain_score ran fill_chain with a CIGAR (score >= min_score of the mate)
    uint64_t ref_pos = 0; int32_t as = 0;
    const uint32_t* cig = nullptr; uint32_t n_cig = 0;
    const uint64_t* alt_pos = nullptr; const int32_t* alt_score = nullptr; uint32_t n_alt = 0;
};
struct PePairOut {
    bool finalized = false;                       // the final paired_chain_score ran
    uint32_t strand = 0;
    int32_t tot = 0, score2 = 0, score2_m[2] = {0, 0}, sub_n = 0, min_score_m[2] = {0, 0};
    PeMateOut mate[2];
};

#define MH_RAW_MAPQ(diff, a) ((int)(6.02 * (diff) / (a) + .499))      // mapq.hpp:144

// mapq.hpp:186-223 (frac_rep == 0: compute_frac_rep returns 0.0, aligner_ksw2.hpp:1979-1981)
static inline void mapq_pe_bwa(int32_t score, int32_t score2, int32_t score_un, int32_t match_score, int32_t sub_n, int32_t score_m1, int32_t score_m2,
                               int32_t score2_m1, int32_t score2_m2, size_t& mapq_m1, size_t& mapq_m2) {
    int32_t mapq = 0;
    const int32_t sub = std::max(score2, score_un);
    mapq = MH_RAW_MAPQ(score - sub, match_score);
    if (sub_n > 0) mapq -= (int)(4.343 * log(sub_n + 1) + .499);
    if (mapq < 0) mapq = 0;
    if (mapq > 60) mapq = 60;
    mapq = (int)(mapq * (1. - .5 * (0. + 0.)) + .499);
    if (score > score_un) {                       // size_t / int32_t mixed as in the reference: every comparison is unsigned
        mapq_m1 = mapq_m1 > (size_t)mapq ? mapq_m1 : (size_t)mapq < mapq_m1 + 40 ? (size_t)mapq : mapq_m1 + 40;
        mapq_m2 = mapq_m2 > (size_t)mapq ? mapq_m2 : (size_t)mapq < mapq_m2 + 40 ? (size_t)mapq : mapq_m2 + 40;
        mapq_m1 = mapq_m1 < (size_t)MH_RAW_MAPQ(score_m1 - score2_m1, match_score) ? mapq_m1 : (size_t)MH_RAW_MAPQ(score_m1 - score2_m1, match_score);
        mapq_m2 = mapq_m2 < (size_t)MH_RAW_MAPQ(score_m2 - score2_m2, match_score) ? mapq_m2 : (size_t)MH_RAW_MAPQ(score_m2 - score2_m2, match_score);
    }
}

static inline void remove_slash_mate(std::string& name) {      // common/sam.hpp:132-141
    const size_t len = name.size();
    if (len >= 2 && name[len - 2] == '/' && (name[len - 1] == '1' || name[len - 1] == '2')) name.resize(len - 2);
}

static inline void sam_write_pe(std::string& out, const Sam& s, const std::string& name, const std::string& seq, const std::string* qual) {
    char buf[32];
    auto d = [&](size_t v) { snprintf(buf, sizeof buf, "%d", (int)v); out += buf; };
    out += name; out.push_back('\t'); d(s.flag); out.push_back('\t'); out += s.rname; out.push_back('\t'); d(s.pos); out.push_back('\t');
    d(s.mapq); out.push_back('\t'); out += s.cigar; out.push_back('\t'); out += s.rnext; out.push_back('\t'); d(s.pnext); out.push_back('\t');
    d((size_t)s.tlen); out.push_back('\t'); out += seq; out.push_back('\t');
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

// the two records of one pair, appended to out.  A: the batch's Aligner (reads / offsets relative to the batch); quals may be null
static inline void pe_emit(const Aligner& A, const moni_align_params_t& P, const PePairOut& R, std::string name1, std::string name2,
                           const uint8_t* reads, const uint8_t* quals, std::string& out) {
    Sam s[2];
    remove_slash_mate(name1); remove_slash_mate(name2);
    if (name1 == name2) { s[0].rnext = "="; s[1].rnext = "="; } else { s[0].rnext = name2; s[1].rnext = name1; }      // aligner_ksw2.hpp:768-777
    bool rev[2] = {false, false};
    if (R.finalized) {
        const uint32_t strand = R.strand;
        rev[0] = strand != 0; rev[1] = strand == 0;          // mate1 / mate2_rev, or mate1_rev / mate2 (aligner_ksw2.hpp:2128-2141)
        bool ok[2] = {false, false};
        for (int k = 0; k < 2; ++k) {
            const PeMateOut& M = R.mate[k];
            if (!M.filled) continue;
            A.finish_record(M.m, M.off, rev[k] ? 1u : 0u, M.ref_pos, M.as, R.score2_m[k], M.cig, M.n_cig, M.alt_pos, M.alt_score, M.n_alt, s[k]);
            s[k].flag = strand ? 16 : 0;                     // chain_score: the pair's strand (aligner_ksw2.hpp:2069)
            s[k].mapq = mapq_se_bwa((int32_t)s[k].as, (int32_t)s[k].zs, (int32_t)s[k].rlen, (int32_t)M.m, (int32_t)P.min_len, P.smatch, P.smismatch, 50.0,
                                    (int32_t)log(50.0f), R.sub_n);
            ok[k] = !s[k].unmapped_lft;
        }
        const size_t l1 = R.mate[0].m, l2 = R.mate[1].m;
        if (ok[0] && ok[1]) {
            s[0].pnext = s[1].pos; s[1].pnext = s[0].pos;
            long long tlen;
            if (s[1].pos > s[0].pos) { tlen = (long long)((s[1].pos + l2) - s[0].pos); s[0].tlen = tlen; s[1].tlen = -tlen; }
            else { tlen = (long long)((s[0].pos + l1) - s[1].pos); s[0].tlen = -tlen; s[1].tlen = tlen; }
            mapq_pe_bwa(R.tot, R.score2, 0, P.smatch, R.sub_n, R.mate[0].score, R.mate[1].score, R.score2_m[0], R.score2_m[1], s[0].mapq, s[1].mapq);
            s[0].as = s[1].as = (size_t)(int64_t)R.tot;
            s[0].zs = s[1].zs = (size_t)(int64_t)R.score2;
            s[0].flag = s[1].flag = 1 | 2;
            if (strand) { s[0].flag |= 16 | 64; s[1].flag |= 32 | 128; }
            else { s[0].flag |= 32 | 64; s[1].flag |= 16 | 128; }
        } else if (ok[0]) {
            s[0].zs = (size_t)(int64_t)R.score2_m[0];
            s[0].flag = 1 | 8 | 64; s[1].flag = 1 | 4 | 128;
            if (strand) s[0].flag |= 16;
            s[1].rname = s[0].rname; s[1].pos = s[0].pos; s[1].mapq = s[0].mapq; s[1].cigar = "*";
            s[1].pnext = s[0].pnext = s[0].pos;
            s[1].tlen = s[0].tlen = 0;
        } else if (ok[1]) {
            s[0].zs = (size_t)(int64_t)R.score2_m[1];        // sic (aligner_ksw2.hpp:2258)
            s[0].flag = 1 | 4 | 64; s[1].flag = 1 | 8 | 128;
            if (!strand) s[1].flag |= 16;
            s[0].rname = s[1].rname; s[0].pos = s[1].pos; s[0].mapq = s[1].mapq; s[0].cigar = "*";
            s[0].pnext = s[1].pnext = s[1].pos;
            s[0].tlen = s[1].tlen = 0;
        } else {
            s[0].flag = s[1].flag = 1 | 4 | 8;
        }
    }
    std::string sq, ql;
    for (int k = 0; k < 2; ++k) {
        const PeMateOut& M = R.mate[k];
        const uint8_t* rd = reads + M.off;
        sq.resize(M.m);
        if (rev[k]) for (uint32_t i = 0; i < M.m; ++i) sq[i] = (char)compl_of(rd[M.m - 1 - i]); else sq.assign((const char*)rd, (const char*)rd + M.m);
        if (quals) { const uint8_t* qv = quals + M.off; ql.resize(M.m); if (rev[k]) for (uint32_t i = 0; i < M.m; ++i) ql[i] = (char)qv[M.m - 1 - i]; else ql.assign((const char*)qv, (const char*)qv + M.m); }
        sam_write_pe(out, s[k], k ? name2 : name1, sq, quals ? &ql : nullptr);
    }
}

// learn_fragment_model (aligner_ksw2.hpp:816-885): Welford over the eligible pairs of one batch, in order, merged into the model
struct PeModel {
    double mean = 0.0, std_dev = 0.0, variance = 0.0, sample_variance = 0.0, m2 = 0.0;
    uint64_t count = 0;
    bool complete = false;
};
// aligned[i]: align(al, false) succeeded; tot / score2 / dist: best_scores[0].tot, best_scores[1].tot, best_scores[0].dist
static inline bool pe_learn_update(PeModel& M, const uint8_t* aligned, const int32_t* tot, const int32_t* score2, const int32_t* min_score, const long long* dist,
                                   uint64_t n, uint64_t learning_n, uint64_t gap_threshold) {
    size_t count = 0; double mean = 0.0, m2acc = 0.0;
    for (uint64_t i = 0; i < n; ++i) {
        if (!aligned[i]) continue;
        const bool second = score2[i] >= min_score[i];
        if (second && !((size_t)(tot[i] - score2[i]) > gap_threshold)) continue;
        const double value = (double)dist[i];
        const double delta = value - mean;
        mean += delta / (++count);
        m2acc += delta * (value - mean);
    }
    const double variance = m2acc / count;
    const double std_dev = sqrt(variance);
    if (!M.complete) {
        if (M.count > 0) {
            const size_t t_count = M.count + count;
            const double delta = M.mean - mean;
            M.m2 += m2acc + (delta * delta * M.count * count) / t_count;
            M.mean = (M.count * M.mean + count * mean) / t_count;
            M.count = t_count;
        } else { M.mean = mean; M.std_dev = std_dev; M.m2 = m2acc; M.count = count; }
        M.complete = M.complete || (M.count >= learning_n);
        if (M.complete) { M.variance = M.m2 / M.count; M.sample_variance = M.m2 / (M.count - 1); M.std_dev = sqrt(M.variance); }
    }
    return M.complete;
}

}  // namespace mh
