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
    bool filled = false;                          // chain_score ran fill_chain with a CIGAR (score >= min_score of the mate), or fill_orphan placed the mate
    bool orphan = false;                          // placed by fill_orphan: its record has no ZS of its own (aligner_ksw2.hpp:2440-2452)
    uint64_t ref_pos = 0; int32_t as = 0;
    const uint32_t* cig = nullptr; uint32_t n_cig = 0;
    const uint64_t* alt_pos = nullptr; const int32_t* alt_score = nullptr; uint32_t n_alt = 0;
};
struct PePairOut {
    bool finalized = false;                       // the final paired_chain_score ran
    uint32_t strand = 0;
    int32_t tot = 0, score2 = 0, score2_m[2] = {0, 0}, sub_n = 0;
    PeMateOut mate[2];
};

static inline int32_t pe_min_score(uint32_t m) { return m ? (int32_t)(20 + 8 * log((double)m)) : INT32_MIN; }      // al.min_score_m1 / _m2 (aligner_ksw2.hpp:1000-1010)

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
            s[k].mapq = mapq_se_bwa((int32_t)s[k].as, R.score2_m[k], (int32_t)s[k].rlen, (int32_t)M.m, (int32_t)P.min_len, P.smatch, P.smismatch, 50.0,
                                    (int32_t)log(50.0f), R.sub_n);
            if (M.orphan) { s[k].zs = 0; s[k].flag = 4; }  // fill_orphan sets neither
            // score.m{1,2}.score >= al.min_score_m{1,2} (aligner_ksw2.hpp:2471,2519): chain_score fills a mate only above its minimum, but
            // fill_orphan places the recovered mate whatever its global score is - below 20 + 8 ln(len) it is reported unmapped
            ok[k] = !s[k].unmapped_lft && (!M.orphan || M.as >= pe_min_score(M.m));
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

// pe_emit in one pass straight into the output text, no per-pair containers (same bytes; tests/host_sim checks one against the other):
// what moni_pe_align_batch's host threads run.  Follows Aligner::emit_record for each mate's lift / MD / NM / CIGAR text.
struct PeMateFin {
    bool filled = false, unmapped_lft = false, cigar_star = true;
    size_t flag = 4, pos = 0, mapq = 255, pnext = 0, as = 0, nm = 0, zs = 0;
    long long tlen = 0;
    int rname = -1;                                   // index into ix.names, -1: "*"
    int lift_rname = -1; size_t lift_pos = 0, lift_nm = 0, rlen = 0;
    size_t md_off = 0, md_len = 0, cs_off = 0, cs_len = 0, os_off = 0, os_len = 0;      // into the scratch: MD text, lifted CIGAR, unlifted CIGAR
};
static inline void pe_emit_fast(const Aligner& A, const moni_align_params_t& P, const PePairOut& R, const char* name1, size_t nl1, const char* name2, size_t nl2,
                                const uint8_t* reads, const uint8_t* quals, std::string& out) {
    const HostIndex& ix = A.ix;
    struct Tabs { uint8_t nt4[256], nt4c[256], cmp[256]; Tabs() { for (int x = 0; x < 256; ++x) { nt4[x] = nt4_of((uint8_t)x); cmp[x] = compl_of((uint8_t)x); nt4c[x] = nt4_of(cmp[x]); } } };
    static const Tabs TB;
    static thread_local std::vector<uint32_t> lcig;
    static thread_local std::vector<char> scratch;
    if (nl1 >= 2 && name1[nl1 - 2] == '/' && (name1[nl1 - 1] == '1' || name1[nl1 - 1] == '2')) nl1 -= 2;      // remove_slash_mate
    if (nl2 >= 2 && name2[nl2 - 2] == '/' && (name2[nl2 - 1] == '1' || name2[nl2 - 1] == '2')) nl2 -= 2;
    const bool same_name = nl1 == nl2 && memcmp(name1, name2, nl1) == 0;
    PeMateFin s[2];
    bool rev[2] = {false, false};
    size_t used = 0;
    if (R.finalized) {
        const uint32_t strand = R.strand;
        rev[0] = strand != 0; rev[1] = strand == 0;
        bool ok[2] = {false, false};
        for (int k = 0; k < 2; ++k) {
            const PeMateOut& M = R.mate[k];
            if (!M.filled) continue;
            PeMateFin& F = s[k];
            F.filled = true;
            const uint32_t m = M.m, qstrand = rev[k] ? 1u : 0u;
            const uint8_t* rd = reads + M.off;
            ix.lift_cigar_at(M.ref_pos, M.cig, M.n_cig, lcig);
            const uint32_t n_lcig = (uint32_t)lcig.size();
            const uint64_t lifted = ix.lift(M.ref_pos);
            uint64_t ref_len = 0, del_len = 0, del_len0 = 0;
            for (uint32_t i = 0; i < n_lcig; ++i) { const uint32_t c = lcig[i]; const int op = c & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += c >> 4; if (op == 2) del_len += c >> 4; }
            for (uint32_t i = 0; i < M.n_cig; ++i) if ((M.cig[i] & 0xf) == 2) del_len0 += M.cig[i] >> 4;
            const size_t need = used + 3 * (size_t)m + 2 * (del_len + del_len0) + 40 * (size_t)(M.n_cig + n_lcig) + 64;
            if (scratch.size() < need) scratch.resize(need + need / 2);
            char* md = scratch.data() + used;
            // nt4 codes by table; the mate's codes once for both passes
            static thread_local std::vector<uint8_t> qcode;
            if (qcode.size() < m) qcode.resize(m + m / 2 + 16);
            if (qstrand) for (uint32_t i = 0; i < m; ++i) qcode[i] = TB.nt4c[rd[m - 1 - i]]; else for (uint32_t i = 0; i < m; ++i) qcode[i] = TB.nt4[rd[i]];
            const uint8_t* const qc = qcode.data();
            const uint8_t* const text = ix.text; const uint64_t n_text = ix.n_text;
            auto tb = [&](uint64_t a) -> uint8_t { return TB.nt4[a < n_text ? text[a] : 0]; };
            auto qb = [&](uint32_t i) -> uint8_t { return qc[i]; };
            auto pass = [&](const uint32_t* cg, uint32_t ncg, uint64_t t, char*& o, bool write) -> int {          // write_MD_core
                int l_MD = 0, nm = 0; uint32_t q = 0;
                for (uint32_t i = 0; i < ncg; ++i) {
                    const int op = cg[i] & 0xf, len = (int)(cg[i] >> 4);
                    if (op == 0 || op == 7 || op == 8) {
                        for (int j = 0; j < len; ++j) {
                            const uint8_t tc = tb(t + j);
                            if (qb(q + j) != tc) { if (write) { o = Aligner::put_int(o, l_MD); *o++ = "ACGTN"[tc]; } l_MD = 0; ++nm; }
                            else ++l_MD;
                        }
                        q += len; t += len;
                    } else if (op == 1) { q += len; nm += len; }
                    else if (op == 2) {
                        if (write) { o = Aligner::put_int(o, l_MD); *o++ = '^'; for (int j = 0; j < len; ++j) *o++ = "ACGTN"[tb(t + j)]; }
                        l_MD = 0; t += len; nm += len;
                    } else if (op == 3) t += len;
                }
                if (write && l_MD > 0) o = Aligner::put_int(o, l_MD);
                return nm;
            };
            F.lift_nm = (size_t)pass(M.cig, M.n_cig, M.ref_pos, md, false);
            F.md_off = (size_t)(md - scratch.data());
            if (ref_len > 0) F.nm = (size_t)pass(lcig.data(), n_lcig, lifted, md, true);
            F.md_len = (size_t)(md - scratch.data()) - F.md_off;
            F.cs_off = (size_t)(md - scratch.data());
            for (uint32_t i = 0; i < n_lcig; ++i) { md = Aligner::put_int(md, (int)(lcig[i] >> 4)); *md++ = "MID"[lcig[i] & 0xf]; }
            F.cs_len = (size_t)(md - scratch.data()) - F.cs_off;
            F.os_off = (size_t)(md - scratch.data());
            for (uint32_t i = 0; i < M.n_cig; ++i) { md = Aligner::put_int(md, (int)(M.cig[i] >> 4)); *md++ = "MID"[M.cig[i] & 0xf]; }
            F.os_len = (size_t)(md - scratch.data()) - F.os_off;
            used = (size_t)(md - scratch.data());
            const auto refi = ix.index(M.ref_pos);
            F.as = (size_t)(int64_t)M.as;
            F.lift_pos = refi.second + 1; F.lift_rname = (int)refi.first;
            if (ref_len > 0) {
                const auto lfti = ix.index(lifted);
                F.pos = lfti.second + 1; F.rname = (int)lfti.first; F.cigar_star = false; F.rlen = ref_len;
            } else { F.pos = 0; F.rname = -1; F.cigar_star = true; F.rlen = 0; F.unmapped_lft = true; F.nm = 0; F.md_len = 0; }
            F.flag = M.orphan ? 4 : (strand ? 16 : 0);
            F.zs = M.orphan ? 0 : (size_t)(int64_t)R.score2_m[k];
            F.mapq = mapq_se_bwa((int32_t)F.as, R.score2_m[k], (int32_t)F.rlen, (int32_t)m, (int32_t)P.min_len, P.smatch, P.smismatch, 50.0, (int32_t)log(50.0f), R.sub_n);
            ok[k] = !F.unmapped_lft && (!M.orphan || M.as >= pe_min_score(M.m));      // as in pe_emit
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
            s[1].rname = s[0].rname; s[1].pos = s[0].pos; s[1].mapq = s[0].mapq; s[1].cigar_star = true;
            s[1].pnext = s[0].pnext = s[0].pos;
            s[1].tlen = s[0].tlen = 0;
        } else if (ok[1]) {
            s[0].zs = (size_t)(int64_t)R.score2_m[1];        // sic (aligner_ksw2.hpp:2258)
            s[0].flag = 1 | 4 | 64; s[1].flag = 1 | 8 | 128;
            if (!strand) s[1].flag |= 16;
            s[0].rname = s[1].rname; s[0].pos = s[1].pos; s[0].mapq = s[1].mapq; s[0].cigar_star = true;
            s[0].pnext = s[1].pnext = s[1].pos;
            s[0].tlen = s[1].tlen = 0;
        } else {
            s[0].flag = s[1].flag = 1 | 4 | 8;
        }
    }
    for (int k = 0; k < 2; ++k) {
        const PeMateOut& M = R.mate[k];
        const PeMateFin& F = s[k];
        const uint32_t m = M.m;
        const uint8_t* rd = reads + M.off;
        const size_t bound = nl1 + nl2 + 2 * (size_t)m + F.md_len + F.cs_len + F.os_len + (size_t)M.n_alt * (A.max_name_len + 26) + 2 * A.max_name_len + 256;
        const size_t at = out.size();
        out.resize(at + bound);
        char* const p0 = &out[at]; char* p = p0;
        p = Aligner::put_str(p, k ? name2 : name1, k ? nl2 : nl1); *p++ = '\t';
        p = Aligner::put_int(p, (int)F.flag); *p++ = '\t';
        if (F.rname >= 0) p = Aligner::put_str(p, ix.names[F.rname]); else *p++ = '*';
        *p++ = '\t'; p = Aligner::put_int(p, (int)F.pos); *p++ = '\t'; p = Aligner::put_int(p, (int)F.mapq); *p++ = '\t';
        if (!F.cigar_star) p = Aligner::put_str(p, scratch.data() + F.cs_off, F.cs_len); else *p++ = '*';
        *p++ = '\t';
        if (same_name) *p++ = '='; else p = Aligner::put_str(p, k ? name1 : name2, k ? nl1 : nl2);
        *p++ = '\t'; p = Aligner::put_int(p, (int)F.pnext); *p++ = '\t'; p = Aligner::put_int(p, (int)(size_t)F.tlen); *p++ = '\t';
        if (rev[k]) { for (uint32_t i = 0; i < m; ++i) p[i] = (char)TB.cmp[rd[m - 1 - i]]; p += m; } else p = Aligner::put_str(p, (const char*)rd, m);
        *p++ = '\t';
        if (quals) { const uint8_t* qv = quals + M.off; if (rev[k]) { for (uint32_t i = 0; i < m; ++i) p[i] = (char)qv[m - 1 - i]; p += m; } else p = Aligner::put_str(p, (const char*)qv, m); }
        else *p++ = '*';
        if (!(F.flag & 4) || F.unmapped_lft) {
            p = Aligner::put_str(p, "\tAS:i:", 6); p = Aligner::put_int(p, (int)F.as); p = Aligner::put_str(p, "\tNM:i:", 6); p = Aligner::put_int(p, (int)F.nm);
            if (F.zs > 0) { p = Aligner::put_str(p, "\tZS:i:", 6); p = Aligner::put_int(p, (int)F.zs); }
            p = Aligner::put_str(p, "\tMD:Z:", 6); p = Aligner::put_str(p, scratch.data() + F.md_off, F.md_len);
            p = Aligner::put_str(p, "\tOA:Z:", 6);
            if (F.lift_rname >= 0) p = Aligner::put_str(p, ix.names[F.lift_rname]); else *p++ = '*';
            *p++ = ','; p = Aligner::put_int(p, (int)F.lift_pos);
            p = Aligner::put_str(p, (F.flag & 16) ? ",-," : ",+,", 3);
            if (F.filled) p = Aligner::put_str(p, scratch.data() + F.os_off, F.os_len); else *p++ = '*';
            *p++ = ','; p = Aligner::put_int(p, (int)F.mapq); *p++ = ','; p = Aligner::put_int(p, (int)F.lift_nm); *p++ = ';';
            p = Aligner::put_str(p, "\tAA:Z:", 6);
            if (F.filled) for (uint32_t i = 0; i < M.n_alt; ++i) {
                const auto r = ix.index(M.alt_pos[i]);
                p = Aligner::put_str(p, ix.names[r.first]); *p++ = ','; p = Aligner::put_int(p, (int)(r.second + 1)); *p++ = ','; p = Aligner::put_int(p, M.alt_score[i]); *p++ = ';';
            }
        }
        *p++ = '\n';
        out.resize(at + (size_t)(p - p0));
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
