// oracle/ - CPU restatement of the reference algorithm; test infrastructure only (tests/, __graft_entry__.smoke(), bench.py's
// cpu_baseline leg).  NOT linked into the product.
//
// align_pe.hpp: the PAIRED-END path of aligner<seed_finder_t> (include/aligner/aligner_ksw2.hpp) restated on top of align.hpp: the checker of
// the product's paired path (moni_pe_learn_batch / moni_pe_align_batch).  Followed, in order:
//   aligner_ksw2.hpp:598-700    orphan_paired_score_t / paired_score_t (operator> used by std::sort)
//   aligner_ksw2.hpp:702-812    paired_alignment_t (min scores, remove_slash_mate, RNEXT)
//   aligner_ksw2.hpp:816-885    learn_fragment_model (Welford, merged across batches)
//   aligner_ksw2.hpp:888-918    align(kpbseq_t*): with find_orphan == false (-u) a pair that fails jointly is written as it stands (the
//                               product's paired path is checked against this mode); with find_orphan == true:
//   aligner_ksw2.hpp:1536-1640  orphan_recovery;  :2329-2560 paired_chain_orphan_score;  :2566-2720 fill_orphan, on klib's ksw_align
//                               (thirdparty/klib, an absent submodule: ksw.c's ksw_i16 + the KSW_XSTART second pass restated below from
//                               the published source as plain DP with its tie rules)
//   aligner_ksw2.hpp:1000-1326  align(paired_alignment_t&, finalize): seeding of the four (mate, strand) patterns with r_offset,
//                               direction filter, frequency filter, chaining, get_best_scores, final paired_chain_score
//   aligner_ksw2.hpp:1329-1431  get_best_scores;  :1471-1534 check_paired_left_MEM;  :2115-2290 paired_chain_score
//   mapq.hpp:186-223            compute_mapq_pe_bwa;  common/sam.hpp:126-142 remove_slash_mate
//   align_reads_dispatcher.hpp:356-389  st_align's paired loop: learn on the first batches, then align them, then the rest
// PARITY UNPINNED: nothing in the reference tree fixes paired-end output.  secondary_chains (-Z) IS restated
// (find_chains_secondary, chain.hpp:442-727, in align.hpp); nothing upstream pins its output either, and it depends
// on std::sort of the chain starts by score alone (ties in libstdc++'s order).  compute_frac_rep returns 0.0 in the reference (aligner_ksw2.hpp:1979-1981) and does here.
#pragma once
#include <mutex>

#include "align.hpp"

namespace oracle {

#define ORC_RAW_MAPQ(diff, a) ((int)(6.02 * (diff) / (a) + .499))      // mapq.hpp:144

// mapq.hpp:186-223
inline size_t compute_mapq_pe_bwa(const int32_t score, const int32_t score2, const int32_t score_un, const int32_t match_score, const int32_t sub_n,
                                  const double frac_rep_m1, const double frac_rep_m2, const int32_t score_m1, const int32_t score_m2,
                                  const int32_t score2_m1, const int32_t score2_m2, size_t& mapq_m1, size_t& mapq_m2) {
    int32_t mapq = 0;
    int32_t sub = std::max(score2, score_un);
    mapq = ORC_RAW_MAPQ(score - sub, match_score);
    if (sub_n > 0) mapq -= (int)(4.343 * log(sub_n + 1) + .499);
    if (mapq < 0) mapq = 0;
    if (mapq > 60) mapq = 60;
    mapq = (int)(mapq * (1. - .5 * (frac_rep_m1 + frac_rep_m2)) + .499);
    if (score > score_un) {
        // the reference mixes size_t and int32_t in these conditionals: the usual arithmetic conversions make every comparison unsigned
        mapq_m1 = mapq_m1 > (size_t)mapq ? mapq_m1 : (size_t)mapq < mapq_m1 + 40 ? (size_t)mapq : mapq_m1 + 40;
        mapq_m2 = mapq_m2 > (size_t)mapq ? mapq_m2 : (size_t)mapq < mapq_m2 + 40 ? (size_t)mapq : mapq_m2 + 40;
        mapq_m1 = mapq_m1 < (size_t)ORC_RAW_MAPQ(score_m1 - score2_m1, match_score) ? mapq_m1 : (size_t)ORC_RAW_MAPQ(score_m1 - score2_m1, match_score);
        mapq_m2 = mapq_m2 < (size_t)ORC_RAW_MAPQ(score_m2 - score2_m2, match_score) ? mapq_m2 : (size_t)ORC_RAW_MAPQ(score_m2 - score2_m2, match_score);
    }
    return mapq;
}

struct pe_config_t {               // aligner_ksw2.hpp:94-128 defaults
    bool filter_dir = true;
    double dir_thr = 50.0;
    size_t ins_learning_n = 1000, ins_learning_score_gap_threshold = 0;
    bool find_orphan = false;          // aligner_ksw2.hpp:128 (true in the reference; -u clears it).  The product is checked in both modes
    bool secondary_chains = false;     // -Z (aligner_ksw2.hpp:126, 1190-1191): find_chains_secondary instead of find_chains
};

struct aligner_pe : aligner {
    pe_config_t pe;
    const int8_t max_penalty;      // aligner_ksw2.hpp:241: max(smatch + smismatch, gapo + gape)
    // insert-size model (aligner_ksw2.hpp:3252-3262)
    double ins_mean = 0.0, ins_std_dev = 0.0, ins_variance = 0.0, ins_sample_variance = 0.0, ins_m2 = 0.0;
    size_t ins_count = 0;
    bool ins_learning_complete = false;
    size_t orphan_pairs = 0, orphan_recovered = 0;      // statistics_t::orphan_reads / orphan_recovered_reads

    aligner_pe(const FlatIndex& ix_, const align_config_t& c, const pe_config_t& p = pe_config_t())
        : aligner(ix_, c), pe(p), max_penalty((int8_t)std::max(c.smatch + c.smismatch, c.gapo + c.gape)) {}

    struct paired_score_t {        // aligner_ksw2.hpp:625-660
        int32_t tot = 0;
        int64_t dist = 0;
        score_t m1, m2;
        size_t chain_i = 0;
        bool paired = false;
    };
    static bool ps_greater(const paired_score_t& lhs, const paired_score_t& rhs) {      // operator> (:649-654)
        return (lhs.tot > rhs.tot) or (lhs.tot == rhs.tot and lhs.m1.lft > rhs.m1.lft) or
               (lhs.tot == rhs.tot and lhs.m1.lft == rhs.m1.lft and lhs.m2.lft > rhs.m2.lft);
    }

    struct paired_alignment_t {    // aligner_ksw2.hpp:662-812
        bool aligned = false, chained = false, best_score = false, second_best_score = false;
        read_t mate1, mate2, mate1_rev, mate2_rev;          // copies: remove_slash_mate edits the names
        sam_t sam_m1, sam_m2;
        int32_t min_score_m1 = 0, min_score_m2 = 0, min_score = 0;
        paired_score_t score;
        int32_t score2 = 0, score2_m1 = 0, score2_m2 = 0;
        float frac_rep_m1 = 0.0, frac_rep_m2 = 0.0;
        int sub_n = 0;
        float mean = 0.0, std_dev = 0.0;
        size_t n_seeds_dir1 = 0, n_seeds_dir2 = 0, n_mems_dir1 = 0, n_mems_dir2 = 0;
        double avg_seed_length_dir1 = 0.0, avg_seed_length_dir2 = 0.0, avg_w_seed_length_dir1 = 0.0, avg_w_seed_length_dir2 = 0.0;
        double armonic_avg_seed_length_dir1 = 0.0, armonic_avg_seed_length_dir2 = 0.0;
        std::vector<mem_t> mems;
        std::vector<std::pair<size_t, size_t>> anchors;
        std::vector<chain_t> chains;
        std::vector<paired_score_t> best_scores;
        csv_t csv_m1;                                       // -c: the pair's MEM statistics (aligner_ksw2.hpp:672, 787-791: one line per pair, under mate 1's name)
    };

    static void remove_slash_mate(read_t& r) {             // common/sam.hpp:132-141
        const size_t len = r.name.size();
        if (len >= 2 and r.name[len - 2] == '/' and (r.name[len - 1] == '1' or r.name[len - 1] == '2')) r.name.resize(len - 2);
    }
    static read_t rc_copy(const read_t& r) {               // rc_copy_kseq_t, kpbseq.h:150-168
        static unsigned char ctab[256]; static bool init = false;
        if (!init) { for (int i = 0; i < 256; ++i) ctab[i] = (unsigned char)i; ctab['A'] = 'T'; ctab['C'] = 'G'; ctab['G'] = 'C'; ctab['T'] = 'A';
                     ctab['a'] = 'T'; ctab['c'] = 'G'; ctab['g'] = 'C'; ctab['t'] = 'A'; init = true; }
        read_t o; o.name = r.name; o.has_qual = r.has_qual;
        const size_t l = r.seq.size();
        o.seq.resize(l);
        for (size_t i = 0; i < l; ++i) o.seq[i] = (char)ctab[(unsigned char)r.seq[l - i - 1]];
        o.qual.assign(r.qual.rbegin(), r.qual.rend());
        return o;
    }
    // paired_alignment_t::init (aligner_ksw2.hpp:748-779)
    void init(paired_alignment_t& al, const read_t& m1, const read_t& m2, double mean_ = 0.0, double std_dev_ = 0.0) {
        al.mate1 = m1; al.mate2 = m2;
        al.min_score_m1 = 20 + 8 * log(al.mate1.seq.size());
        al.min_score_m2 = 20 + 8 * log(al.mate2.seq.size());
        al.min_score = al.min_score_m1 + al.min_score_m2;
        al.mean = mean_; al.std_dev = std_dev_;
        remove_slash_mate(al.mate1); remove_slash_mate(al.mate2);
        al.mate1_rev = rc_copy(al.mate1); al.mate2_rev = rc_copy(al.mate2);
        al.sam_m1.read = &al.mate1; al.sam_m2.read = &al.mate2;
        if (al.mate1.name == al.mate2.name) { al.sam_m1.rnext = "="; al.sam_m2.rnext = "="; }
        else { al.sam_m1.rnext = al.mate2.name; al.sam_m2.rnext = al.mate1.name; }
    }

    // aligner_ksw2.hpp:1471-1534.  A chain without anchors of one of the mates reads that mate's coordinate uninitialised in the
    // reference; such a chain scores 0 in paired_chain_score whatever happens here, so the value (0) never reaches the output.
    bool check_paired_left_MEM(std::vector<std::pair<size_t, size_t>>& m1_vec, std::vector<std::pair<size_t, size_t>>& m2_vec, paired_alignment_t& al, size_t i) {
        auto& chain = al.chains[i];
        chain.reverse();
        size_t m1_ref = 0, m2_ref = 0;
        for (size_t j = 0; j < chain.anchors.size(); ++j) {
            const size_t a = chain.anchors[j];
            if ((al.mems[al.anchors[a].first].mate & MATE_2) == 0) { m1_ref = ix.index(ix.lift(al.mems[al.anchors[a].first].occs[al.anchors[a].second])).second + 1; break; }
        }
        for (size_t j = 0; j < chain.anchors.size(); ++j) {
            const size_t a = chain.anchors[j];
            if ((al.mems[al.anchors[a].first].mate & MATE_2) != 0) { m2_ref = ix.index(ix.lift(al.mems[al.anchors[a].first].occs[al.anchors[a].second])).second + 1; break; }
        }
        bool discovered = false;
        for (size_t j = 0; j < m1_vec.size(); ++j)
            if ((ORC_DIST(m1_vec[j].first, m1_ref) < cfg.region_dist) and (ORC_DIST(m2_vec[j].first, m2_ref) < cfg.region_dist))
                if (m1_vec[j].second == (size_t)al.chains[i].score) discovered = true;
        chain.reset();
        if (discovered) return true;
        m1_vec.push_back(std::make_pair(m1_ref, (size_t)al.chains[i].score));
        m2_vec.push_back(std::make_pair(m2_ref, (size_t)al.chains[i].score));
        return false;
    }

    // the pairing term of aligner_ksw2.hpp:2176-2181 / 2193-2198
    int32_t pair_total(const paired_score_t& s, const paired_alignment_t& al) const {
        double ns = 0.0;
        if (al.std_dev > 0.0) ns = (s.dist - al.mean) / al.std_dev;
        int32_t tot = (int)(s.m1.score + s.m2.score + .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * cfg.smatch + .499);
        if (tot < 0) tot = 0;
        return tot;
    }

    // the tail of paired_chain_score / paired_chain_orphan_score (aligner_ksw2.hpp:2200-2288, 2470-2556): PNEXT / TLEN / flags / paired MAPQ,
    // or one mate placed by the other
    void pair_tail(paired_alignment_t& al, const int32_t tot, const score_t& m1, const score_t& m2, const uint8_t strand, const read_t* mate1, const read_t* mate2) {
        sam_t& sam_m1 = al.sam_m1; sam_t& sam_m2 = al.sam_m2;
        if (m1.score >= al.min_score_m1 && !m1.unmapped_lft && m2.score >= al.min_score_m2 && !m2.unmapped_lft) {
            sam_m1.pnext = sam_m2.pos; sam_m2.pnext = sam_m1.pos;
            ll tlen;
            if (sam_m2.pos > sam_m1.pos) { tlen = (sam_m2.pos + mate2->seq.size()) - sam_m1.pos; sam_m1.tlen = tlen; sam_m2.tlen = -tlen; }
            else { tlen = (sam_m1.pos + mate1->seq.size()) - sam_m2.pos; sam_m1.tlen = -tlen; sam_m2.tlen = tlen; }
            const int32_t score_un = 0;
            compute_mapq_pe_bwa(tot, al.score2, score_un, cfg.smatch, al.sub_n, al.frac_rep_m1, al.frac_rep_m2, m1.score, m2.score,
                                al.score2_m1, al.score2_m2, sam_m1.mapq, sam_m2.mapq);
            sam_m1.as = tot; sam_m2.as = tot;
            sam_m1.zs = al.score2; sam_m2.zs = al.score2;
            sam_m1.flag = sam_m2.flag = 1 | 2;                                  // SAM_PAIRED | SAM_MAPPED_PAIRED
            if (strand) { sam_m1.flag |= 16 | 64; sam_m2.flag |= 32 | 128; }    // REVERSED | FIRST ; MATE_REVERSED | SECOND
            else { sam_m1.flag |= 32 | 64; sam_m2.flag |= 16 | 128; }
        } else if (m1.score >= al.min_score_m1 && !m1.unmapped_lft) {
            sam_m1.zs = al.score2_m1;
            sam_m1.flag = 1 | 8 | 64;                                           // PAIRED | MATE_UNMAPPED | FIRST
            sam_m2.flag = 1 | 4 | 128;                                          // PAIRED | UNMAPPED | SECOND
            if (strand) sam_m1.flag |= 16;
            sam_m2.rname = sam_m1.rname; sam_m2.pos = sam_m1.pos; sam_m2.mapq = sam_m1.mapq; sam_m2.cigar = "*";
            sam_m2.pnext = sam_m1.pnext = sam_m1.pos;
            sam_m2.tlen = sam_m1.tlen = 0;
        } else if (m2.score >= al.min_score_m2 && !m2.unmapped_lft) {
            sam_m1.zs = al.score2_m2;                                           // sic (aligner_ksw2.hpp:2258)
            sam_m1.flag = 1 | 4 | 64;
            sam_m2.flag = 1 | 8 | 128;
            if (not strand) sam_m2.flag |= 16;
            sam_m1.rname = sam_m2.rname; sam_m1.pos = sam_m2.pos; sam_m1.mapq = sam_m2.mapq; sam_m1.cigar = "*";
            sam_m1.pnext = sam_m2.pnext = sam_m2.pos;
            sam_m1.tlen = sam_m2.tlen = 0;
        } else {
            sam_m1.flag = sam_m2.flag = 1 | 4 | 8;
        }
    }

    // aligner_ksw2.hpp:2115-2290
    paired_score_t paired_chain_score(paired_alignment_t& al, const size_t chain_i, const bool score_only = true) {
        auto& chain = al.chains[chain_i];
        chain.reverse();                                   // lazily, and left reversed (as in the reference)
        const read_t* mate1; const read_t* mate2;
        uint8_t strand = 0;
        if ((chain.mate == 0) || ((chain.mate & MATE_RC) and (chain.mate & MATE_2))) { mate1 = &al.mate1; mate2 = &al.mate2_rev; }
        else { mate1 = &al.mate1_rev; mate2 = &al.mate2; strand = 1; }
        paired_score_t score;
        score.chain_i = chain_i;
        if (!chain.paired) return score;
        std::vector<size_t> c1, c2;
        for (size_t i = 0; i < chain.anchors.size(); ++i) {
            const size_t a = chain.anchors[i];
            if ((al.mems[al.anchors[a].first].mate & MATE_2) == 0) c1.push_back(a); else c2.push_back(a);
        }
        score.paired = chain.paired;
        sam_t& sam_m1 = al.sam_m1; sam_t& sam_m2 = al.sam_m2;
        if (score_only) {
            score.m1 = chain_score(c1, al.anchors, al.mems, al.min_score_m1, mate1);
            score.m2 = chain_score(c2, al.anchors, al.mems, al.min_score_m2, mate2);
        } else {
            score.m1 = chain_score(c1, al.anchors, al.mems, al.min_score_m1, mate1, false, al.score2_m1, strand, &sam_m1, al.sub_n, al.frac_rep_m1);
            score.m2 = chain_score(c2, al.anchors, al.mems, al.min_score_m2, mate2, false, al.score2_m2, strand, &sam_m2, al.sub_n, al.frac_rep_m2);
        }
        score.dist = (int64_t)ORC_DIST(score.m2.pos, (score.m1.pos + mate1->seq.size()));
        score.tot = pair_total(score, al);
        score.m1.lft = ix.lift(score.m1.pos);
        score.m2.lft = ix.lift(score.m2.pos);
        if (score_only) return score;
        sam_m1.read = mate1; sam_m2.read = mate2;
        pair_tail(al, score.tot, score.m1, score.m2, strand, mate1, mate2);
        return score;
    }

    // aligner_ksw2.hpp:1329-1431
    void get_best_scores(paired_alignment_t& al, size_t k) {
        std::set<size_t> different_scores;
        size_t i = 0;
        std::vector<std::pair<size_t, size_t>> m1_left, m2_left;
        int32_t m1_max = 0, m2_max = 0;
        std::vector<std::string> m1_h, m2_h; std::vector<size_t> m1_p, m1_s, m2_p, m2_s;
        while (i < al.chains.size() and different_scores.size() < k) {
            different_scores.insert(al.chains[i].score);
            if (cfg.left_mem_check) {
                if (check_paired_left_MEM(m1_left, m2_left, al, i)) { ++i; al.csv_m1.num_chains_skipped++; continue; }      // aligner_ksw2.hpp:1354-1358
            }
            if (different_scores.size() < k) {
                paired_score_t score = paired_chain_score(al, i);
                m1_max = check_max_score(m1_max, score.m1, m1_h, m1_p, m1_s);
                m2_max = check_max_score(m2_max, score.m2, m2_h, m2_p, m2_s);
                if (score.tot >= al.min_score) {
                    bool replaced = false;
                    for (size_t j = 0; j < al.best_scores.size(); ++j) {
                        paired_score_t zero; zero.chain_i = i;
                        if ((ORC_DIST(al.best_scores[j].m1.lft, score.m1.lft) < cfg.region_dist) and (ORC_DIST(al.best_scores[j].m2.lft, score.m2.lft) < cfg.region_dist)) {
                            if (score.tot > al.best_scores[j].tot) {
                                if (replaced) al.best_scores[j] = zero;
                                else { al.best_scores[j] = score; replaced = true; }
                            } else if (score.tot <= al.best_scores[j].tot) { j = al.best_scores.size(); replaced = true; }
                        }
                    }
                    if (not replaced) al.best_scores.push_back(score);
                }
                ++i;
            }
        }
        al.sam_m1.alt_haplotypes = m1_h; al.sam_m1.alt_pos = m1_p; al.sam_m1.alt_scores = m1_s;
        al.sam_m2.alt_haplotypes = m2_h; al.sam_m2.alt_pos = m2_p; al.sam_m2.alt_scores = m2_s;
        paired_score_t zero; zero.chain_i = al.chains.size();
        while (al.best_scores.size() < 2) al.best_scores.push_back(zero);
        std::sort(al.best_scores.begin(), al.best_scores.end(), ps_greater);
        size_t j = 1;
        al.sub_n = 0;
        while (j < al.best_scores.size() and al.best_scores[j++].tot >= (al.best_scores[0].tot - max_penalty)) ++al.sub_n;
        al.best_score = true;
        al.score2 = al.best_scores[1].tot; al.score2_m1 = al.best_scores[1].m1.score; al.score2_m2 = al.best_scores[1].m2.score;
        al.second_best_score = (al.score2 >= al.min_score);
    }

    // aligner_ksw2.hpp:1000-1326 (secondary_chains: find_chains_secondary below); mems_out: the `out` of a call with report_mems (learn_fragment_model passes none)
    bool align(paired_alignment_t& al, bool finalize = true, std::string* mems_out = nullptr) {
        const size_t l1 = al.mate1.seq.size(), l2 = al.mate2.seq.size();
        if (pe.filter_dir) {
            mem_finder.find_mems(al.mate1.seq.data(), l1, al.mems, 0, MATE_1 | MATE_F);
            mem_finder.find_mems(al.mate2_rev.seq.data(), l2, al.mems, l1, MATE_2 | MATE_RC);
            al.n_mems_dir1 = al.mems.size(); al.n_seeds_dir1 = 0;
            mem_finder.find_mems(al.mate2.seq.data(), l2, al.mems, 0, MATE_2 | MATE_F);
            mem_finder.find_mems(al.mate1_rev.seq.data(), l1, al.mems, l2, MATE_1 | MATE_RC);
            al.n_mems_dir2 = al.mems.size() - al.n_mems_dir1; al.n_seeds_dir2 = 0;
            mem_finder.populate_seeds(al.mems, cfg.report_mems);
            if (csv_out) calculate_MEM_stats(al.mems, al.csv_m1);          // aligner_ksw2.hpp:1030-1031
            // NB: populate_seeds appends the halves of long MEMs behind all four calls' MEMs; the direction statistics below run over
            // the first n_mems_dir1 entries and "the rest", halves included, exactly as the reference's index arithmetic does
            for (size_t i = 0; i < al.n_mems_dir1; ++i) {
                al.n_seeds_dir1 += al.mems[i].occs.size();
                al.avg_seed_length_dir1 += al.mems[i].len;
                al.avg_w_seed_length_dir1 += al.mems[i].len * al.mems[i].occs.size();
                al.armonic_avg_seed_length_dir1 += (double)l1 / (double)al.mems[i].len;
            }
            if (al.n_mems_dir1 > 0) { al.avg_seed_length_dir1 /= al.n_mems_dir1; al.avg_w_seed_length_dir1 /= al.n_seeds_dir1; al.armonic_avg_seed_length_dir1 = (double)al.n_mems_dir1 / al.armonic_avg_seed_length_dir1; }
            for (size_t i = al.n_mems_dir1; i < al.mems.size(); ++i) {
                al.n_seeds_dir2 += al.mems[i].occs.size();
                al.avg_seed_length_dir2 += al.mems[i].len;
                al.avg_w_seed_length_dir2 += al.mems[i].len * al.mems[i].occs.size();
                al.armonic_avg_seed_length_dir2 += (double)l2 / (double)al.mems[i].len;
            }
            if (al.n_mems_dir2 > 0) { al.avg_seed_length_dir2 /= al.n_mems_dir2; al.avg_w_seed_length_dir2 /= al.n_seeds_dir2; al.armonic_avg_seed_length_dir2 = (double)al.n_mems_dir2 / al.armonic_avg_seed_length_dir2; }
            if ((al.avg_seed_length_dir1 > al.avg_seed_length_dir2) and ((al.avg_seed_length_dir1 - al.avg_seed_length_dir2) > pe.dir_thr)) {
                for (size_t i = al.n_mems_dir1; i < al.mems.size(); ++i) al.csv_m1.num_mems_filter += al.mems[i].occs.size();      // aligner_ksw2.hpp:1066-1068
                al.mems.erase(al.mems.begin() + al.n_mems_dir1, al.mems.end());
            }
            if ((al.avg_seed_length_dir2 > al.avg_seed_length_dir1) and ((al.avg_seed_length_dir2 - al.avg_seed_length_dir1) > pe.dir_thr)) {
                for (size_t i = 0; i < al.n_mems_dir1; ++i) al.csv_m1.num_mems_filter += al.mems[i].occs.size();                  // aligner_ksw2.hpp:1073-1075
                al.mems.erase(al.mems.begin(), al.mems.begin() + al.n_mems_dir1);
            }
            if (cfg.filter_freq) seed_freq_filter(al.mems, cfg.freq_thr, al.csv_m1);
        } else {
            mem_finder.find_mems(al.mate1.seq.data(), l1, al.mems, 0, MATE_1 | MATE_F);
            mem_finder.find_mems(al.mate1_rev.seq.data(), l1, al.mems, l2, MATE_1 | MATE_RC);
            mem_finder.find_mems(al.mate2.seq.data(), l2, al.mems, 0, MATE_2 | MATE_F);
            mem_finder.find_mems(al.mate2_rev.seq.data(), l2, al.mems, l1, MATE_2 | MATE_RC);
            mem_finder.populate_seeds(al.mems, cfg.report_mems);
            if (csv_out) calculate_MEM_stats(al.mems, al.csv_m1);          // aligner_ksw2.hpp:1115-1116
            if (cfg.filter_freq) seed_freq_filter(al.mems, cfg.freq_thr, al.csv_m1);
        }
        if (cfg.report_mems && mems_out != nullptr) {               // aligner_ksw2.hpp:1118-1180: one secondary record per occurrence of every MEM left
            for (size_t i = 0; i < al.mems.size(); ++i) {
                const bool is_m1 = al.mems[i].mate == (MATE_1 | MATE_F) || al.mems[i].mate == (MATE_1 | MATE_RC);
                const read_t& src = is_m1 ? ((al.mems[i].mate & MATE_RC) ? al.mate1_rev : al.mate1) : ((al.mems[i].mate & MATE_RC) ? al.mate2_rev : al.mate2);
                read_t part;                                         // copy_partial_kseq_t (kpbseq.h:197-205)
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
                    write_sam(*mems_out, rs);
                }
            }
            al.aligned = true;
            return true;
        }
        al.frac_rep_m1 = 0.0; al.frac_rep_m2 = 0.0;                  // compute_frac_rep (aligner_ksw2.hpp:1973-1981) returns 0.0
        al.chained = pe.secondary_chains ? find_chains_secondary(al.mems, al.anchors, al.chains, cfg.chain) : find_chains(al.mems, al.anchors, al.chains, cfg.chain);      // aligner_ksw2.hpp:1190-1194
        if (not al.chained) return false;
        get_best_scores(al, cfg.check_k);
        auto& best = al.best_scores;
        if (best[0].tot < al.min_score) {
            al.sam_m1.alt_haplotypes.clear(); al.sam_m1.alt_pos.clear(); al.sam_m1.alt_scores.clear();
            al.sam_m2.alt_haplotypes.clear(); al.sam_m2.alt_pos.clear(); al.sam_m2.alt_scores.clear();
            return false;
        }
        if (finalize) {
            al.score = paired_chain_score(al, best[0].chain_i, false);
            al.aligned = (al.score.tot >= al.min_score);
        } else al.aligned = (best[0].tot >= al.min_score);
        return al.aligned;
    }

    // ---- klib ksw.c (attractivechaos/klib; the submodule is empty): kswr_t of ksw_align(.., xtra = KSW_XSTART) through ksw_i16 -------------
    // Local alignment, rows = target.  H = max(0, diag + s, E, F); E / F lose gape per step and restart from H - (gapo + gape), never below 0
    // (unsigned saturating subtraction).  te: the first row whose maximum exceeds every earlier row's; qe: the smallest query index holding
    // that row's maximum.  The second pass aligns the reversed prefixes and stops at the first row that reaches the score: tb / qb.
    struct kswr_t { int score = 0, te = -1, qe = -1, score2 = -1, te2 = -1, tb = -1, qb = -1; };
    static kswr_t ksw_pass(int qlen, const uint8_t* q, int tlen, const uint8_t* t, int m_, const int8_t* mat_, int gapo_, int gape_, int endsc) {
        kswr_t r;
        std::vector<int> H0(qlen + 1, 0), H1(qlen + 1, 0), E(qlen + 1, 0), Hmax(qlen + 1, 0);
        const int gapoe = gapo_ + gape_;
        int gmax = 0, te = -1;
        for (int i = 0; i < tlen; ++i) {
            int f = 0, imax = 0;
            const int8_t* row = mat_ + (int)t[i] * m_;
            for (int j = 0; j < qlen; ++j) {
                int h = (j ? H0[j - 1] : 0) + row[q[j]];
                h = std::max(h, E[j]); h = std::max(h, f);
                H1[j] = h;
                imax = std::max(imax, h);
                const int hh = std::max(h - gapoe, 0);
                E[j] = std::max(std::max(E[j] - gape_, 0), hh);
                f = std::max(std::max(f - gape_, 0), hh);
            }
            if (imax > gmax) { gmax = imax; te = i; Hmax = H1; if (gmax >= endsc) break; }
            std::swap(H0, H1);
        }
        r.score = gmax; r.te = te;
        int mx = -1;
        for (int j = 0; j < qlen; ++j) if (Hmax[j] > mx) { mx = Hmax[j]; r.qe = j; }
        return r;
    }
    static kswr_t ksw_align(int qlen, uint8_t* query, int tlen, uint8_t* target, int m_, const int8_t* mat_, int gapo_, int gape_) {
        kswr_t r = ksw_pass(qlen, query, tlen, target, m_, mat_, gapo_, gape_, 0x10000);
        std::reverse(query, query + (r.qe + 1)); std::reverse(target, target + (r.te + 1));
        const kswr_t rr = ksw_pass(r.qe + 1, query, tlen, target, m_, mat_, gapo_, gape_, r.score);
        std::reverse(query, query + (r.qe + 1)); std::reverse(target, target + (r.te + 1));
        if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
        return r;
    }

    struct orphan_paired_score_t {     // aligner_ksw2.hpp:599-622
        int32_t tot = 0;
        int64_t dist = 0;
        score_t m1, m2;
        size_t chain_i = 0;
        std::pair<size_t, size_t> pos = std::make_pair(0, 0);
    };
    static bool ops_greater(const orphan_paired_score_t& lhs, const orphan_paired_score_t& rhs) {
        return (lhs.tot > rhs.tot) or (lhs.tot == rhs.tot and lhs.m1.lft > rhs.m1.lft) or
               (lhs.tot == rhs.tot and lhs.m1.lft == rhs.m1.lft and lhs.m2.lft > rhs.m2.lft);
    }

    // aligner_ksw2.hpp:2566-2720
    score_t fill_orphan(ll& start, ll& end, const read_t* paired_mate, const bool score_only = true, sam_t* sam = nullptr) {
        score_t score;
        const size_t ref_occ = start;
        ll ref_len = end - start + 1;
        std::vector<uint8_t> refv(ref_len + 1);
        uint8_t* ref = refv.data();
        expand_nt4(ref_occ, ref_len, ref);
        const size_t seq_len = paired_mate->seq.size();
        std::vector<uint8_t> seqv(seq_len + 1);
        uint8_t* seq = seqv.data();
        for (size_t i = 0; i < seq_len; ++i) seq[i] = seq_nt4_table[(unsigned char)paired_mate->seq[i]];
        if (score_only) {
            const kswr_t r = ksw_align((int)seq_len, seq, (int)ref_len, ref, 5, mat, cfg.gapo, cfg.gape);
            end = start + r.te;
            start += r.tb;
            const size_t ref_len_ = r.te - r.tb + 1;
            ksw_extz_t ez;
            memset(&ez, 0, sizeof(ksw_extz_t));
            ksw_reset_extz(&ez);
            if (r.tb >= 0) extz(seq_len, seq, ref_len_, ref + r.tb, KSW_EZ_SCORE_ONLY, &ez);      // (tb == -1 reads before the buffer in the reference)
            score.score = ez.score;
            score.pos = start;
            if (not ix.valid(start, end - start + 1)) score.score = std::numeric_limits<int32_t>::min();
            return score;
        }
        ksw_extz_t ez;
        memset(&ez, 0, sizeof(ksw_extz_t));
        ksw_reset_extz(&ez);
        extz(seq_len, seq, ref_len, ref, KSW_EZ_RIGHT, &ez);
        sam->lift_cigar = "";
        for (int i = 0; i < ez.n_cigar; ++i) sam->lift_cigar += std::to_string(ez.cigar[i] >> 4) + "MID"[ez.cigar[i] & 0xf];
        sam->lift_nm = write_MD_core(ref, seq, ez.cigar, ez.n_cigar, sam->lift_md);
        const auto refi = ix.index(ref_occ);
        sam->as = ez.score;
        sam->lift_pos = refi.second + 1;
        sam->lift_rname = ix.names[refi.first];
        sam->lift_rlen = ref_len;
        const std::vector<uint32_t> lft_cigar = ix.lift_cigar(ez.cigar, ez.n_cigar, ref_occ);
        const auto lift = ix.lift(ref_occ);
        const auto lft_ref = ix.index(lift);
        sam->pos = lft_ref.second + 1;
        sam->rname = ix.names[lft_ref.first];
        sam->cigar = "";
        for (size_t i = 0; i < lft_cigar.size(); ++i) sam->cigar += std::to_string(lft_cigar[i] >> 4) + "MID"[lft_cigar[i] & 0xf];
        size_t l_len = 0;
        for (size_t i = 0; i < lft_cigar.size(); ++i) { int op = lft_cigar[i] & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) l_len += lft_cigar[i] >> 4; }
        if (l_len > 0) {
            std::vector<uint8_t> l_ref(l_len + 1);
            expand_nt4(lift, l_len, l_ref.data());
            sam->nm = write_MD_core(l_ref.data(), seq, lft_cigar.data(), lft_cigar.size(), sam->md);
            sam->rlen = l_len;
            score.score = ez.score;
            score.pos = start;
        } else {
            sam->pos = 0; sam->rname = "*"; sam->cigar = "*"; sam->rlen = 0;
            sam->unmapped_lft = true;
            score.unmapped_lft = true;
        }
        free(ez.cigar);
        return score;
    }

    // aligner_ksw2.hpp:2329-2560
    orphan_paired_score_t paired_chain_orphan_score(paired_alignment_t& al, const size_t chain_i, const double mean, const double std_dev,
                                                    const bool score_only = true, ll start = 0, ll end = 0) {
        auto& chain = al.chains[chain_i];
        chain.reverse();
        const read_t* mate1; const read_t* mate2;
        uint8_t strand = 0;
        if ((chain.mate == 0) || ((chain.mate & MATE_RC) and (chain.mate & MATE_2))) { mate1 = &al.mate1; mate2 = &al.mate2_rev; }
        else { mate1 = &al.mate1_rev; mate2 = &al.mate2; strand = 1; }
        orphan_paired_score_t score;
        score.chain_i = chain_i;
        std::vector<size_t> c1, c2;
        size_t lm_pos = (size_t)-1, rm_pos = 0;
        for (size_t i = 0; i < chain.anchors.size(); ++i) {
            const size_t a = chain.anchors[i];
            const mem_t& mem = al.mems[al.anchors[a].first];
            rm_pos = std::max(rm_pos, mem.occs[al.anchors[a].second] + mem.len);
            lm_pos = std::min(lm_pos, mem.occs[al.anchors[a].second]);
            if ((mem.mate & MATE_2) == 0) c1.push_back(a); else c2.push_back(a);
        }
        sam_t& sam_m1 = al.sam_m1; sam_t& sam_m2 = al.sam_m2;
        const ll lim = (ll)(n - ix.w);
        if (score_only) {
            if (c1.size() > 0) {
                score.m1 = chain_score(c1, al.anchors, al.mems, al.min_score_m1, mate1);
                start = rm_pos + (ll)std::floor(mean - 4 * std_dev);
                end = rm_pos + (ll)std::ceil(mean + 4 * std_dev);
                start = std::max(start, (ll)0); start = std::min(start, lim); end = std::min(end, lim);
                if (start < end) score.m2 = fill_orphan(start, end, mate2);
                score.pos = std::make_pair((size_t)start, (size_t)end);
            } else {
                score.m2 = chain_score(c2, al.anchors, al.mems, al.min_score_m2, mate2);
                start = lm_pos + (ll)std::floor(-mean - 4 * std_dev);
                end = lm_pos + (ll)std::ceil(-mean + 4 * std_dev);
                start = std::max(start, (ll)0); start = std::min(start, lim); end = std::min(end, lim);
                if (start < end) score.m1 = fill_orphan(start, end, mate1);
                score.pos = std::make_pair((size_t)start, (size_t)end);
            }
        } else {
            if (c1.size() > 0) {
                score.m1 = chain_score(c1, al.anchors, al.mems, al.min_score_m1, mate1, false, al.score2_m1, strand, &sam_m1, al.sub_n, al.frac_rep_m1);
                if (start < end) score.m2 = fill_orphan(start, end, mate2, false, &sam_m2);
                sam_m2.mapq = compute_mapq_se_bwa(sam_m2.as, al.score2_m2, sam_m2.rlen, mate2->seq.size(), cfg.min_len, cfg.smatch, cfg.smismatch,
                                                  mapq_coeff_len, mapq_coeff_fac, al.sub_n, 0, al.frac_rep_m2);
            } else {
                if (start < end) score.m1 = fill_orphan(start, end, mate1, false, &sam_m1);
                score.m2 = chain_score(c2, al.anchors, al.mems, al.min_score_m2, mate2, false, al.score2_m2, strand, &sam_m2, al.sub_n, al.frac_rep_m2);
                sam_m1.mapq = compute_mapq_se_bwa(sam_m1.as, al.score2_m1, sam_m1.rlen, mate1->seq.size(), cfg.min_len, cfg.smatch, cfg.smismatch,
                                                  mapq_coeff_len, mapq_coeff_fac, al.sub_n, 0, al.frac_rep_m1);
            }
        }
        score.dist = (int64_t)ORC_DIST(score.m2.pos, (score.m1.pos + mate1->seq.size()));
        {
            paired_score_t tmp; tmp.dist = score.dist; tmp.m1 = score.m1; tmp.m2 = score.m2;
            score.tot = pair_total(tmp, al);
        }
        score.m1.lft = ix.lift(score.m1.pos);
        score.m2.lft = ix.lift(score.m2.pos);
        if (score_only) return score;
        sam_m1.read = mate1; sam_m2.read = mate2;
        pair_tail(al, score.tot, score.m1, score.m2, strand, mate1, mate2);
        return score;
    }

    // aligner_ksw2.hpp:1536-1640
    bool orphan_recovery(paired_alignment_t& al, const double mean, const double std_dev) {
        std::vector<orphan_paired_score_t> best_scores;
        for (size_t i = 0; i < al.chains.size(); ++i) {
            orphan_paired_score_t score = paired_chain_orphan_score(al, i, mean, std_dev);
            if (score.tot >= al.min_score) {
                bool replaced = false;
                for (size_t j = 0; j < best_scores.size(); ++j) {
                    orphan_paired_score_t zero; zero.chain_i = i;
                    if ((ORC_DIST(best_scores[j].m1.lft, score.m1.lft) < cfg.region_dist) and (ORC_DIST(best_scores[j].m2.lft, score.m2.lft) < cfg.region_dist)) {
                        if (score.tot > best_scores[j].tot) {
                            if (replaced) best_scores[j] = zero;
                            else { best_scores[j] = score; replaced = true; }
                        } else if (score.tot <= best_scores[j].tot) { j = best_scores.size(); replaced = true; }
                    }
                }
                if (not replaced) best_scores.push_back(score);
            }
        }
        orphan_paired_score_t zero; zero.chain_i = al.chains.size();
        while (best_scores.size() < 2) best_scores.push_back(zero);
        std::sort(best_scores.begin(), best_scores.end(), ops_greater);
        if (best_scores[0].tot < al.min_score) return false;
        size_t j = 1;
        al.sub_n = 0;
        while (j < best_scores.size() and best_scores[j++].tot >= (best_scores[0].tot - max_penalty)) ++al.sub_n;
        al.best_score = true;
        al.score2 = best_scores[1].tot; al.score2_m1 = best_scores[1].m1.score; al.score2_m2 = best_scores[1].m2.score;
        al.second_best_score = (al.score2 >= al.min_score);
        const orphan_paired_score_t fin = paired_chain_orphan_score(al, best_scores[0].chain_i, mean, std_dev, false, (ll)best_scores[0].pos.first, (ll)best_scores[0].pos.second);
        al.score.tot = fin.tot; al.score.dist = fin.dist; al.score.m1 = fin.m1; al.score.m2 = fin.m2; al.score.chain_i = fin.chain_i;
        al.aligned = (al.score.tot >= al.min_score);
        return al.aligned;
    }

    // aligner_ksw2.hpp:816-885 (one thread: no mutex needed)
    bool learn_fragment_model(const std::vector<read_t>& m1, const std::vector<read_t>& m2) {
        size_t count = 0; double mean = 0.0, m2acc = 0.0;
        for (size_t i = 0; i < m1.size(); ++i) {
            paired_alignment_t al;
            init(al, m1[i], m2[i]);
            if (align(al, false) and ((not al.second_best_score) or ((size_t)(al.best_scores[0].tot - al.best_scores[1].tot) > pe.ins_learning_score_gap_threshold))) {
                const double value = (double)(al.best_scores[0].dist);
                const double delta = value - mean;
                mean += delta / (++count);
                m2acc += delta * (value - mean);
            }
        }
        const double variance = m2acc / count;
        const double std_dev = sqrt(variance);
        if (not ins_learning_complete) {
            if (ins_count > 0) {
                const size_t t_count = ins_count + count;
                const double delta = ins_mean - mean;
                ins_m2 += m2acc + (delta * delta * ins_count * count) / t_count;
                ins_mean = (ins_count * ins_mean + count * mean) / t_count;
                ins_count = t_count;
            } else { ins_mean = mean; ins_std_dev = std_dev; ins_m2 = m2acc; ins_count = count; }
            ins_learning_complete = ins_learning_complete or (ins_count >= pe.ins_learning_n);
            if (ins_learning_complete) { ins_variance = ins_m2 / ins_count; ins_sample_variance = ins_m2 / (ins_count - 1); ins_std_dev = sqrt(ins_variance); }
        }
        return ins_learning_complete;
    }

    // aligner_ksw2.hpp:888-918: both records of every pair, in order
    size_t align_batch(const std::vector<read_t>& m1, const std::vector<read_t>& m2, std::string& out) {
        size_t aligned = 0;
        for (size_t i = 0; i < m1.size(); ++i) {
            paired_alignment_t al;
            init(al, m1[i], m2[i]);
            al.mean = ins_mean; al.std_dev = ins_std_dev;
            if (not align(al, true, &out) and al.chained) {        // aligner_ksw2.hpp:900-906
                ++orphan_pairs;
                if (pe.find_orphan) orphan_recovery(al, ins_mean, ins_std_dev);
                if (al.aligned) ++orphan_recovered;
            }
            if (!cfg.report_mems) {                                  // (report_mems: align wrote the MEM records itself)
                write_sam(out, al.sam_m1);
                write_sam(out, al.sam_m2);
            }
            if (csv_out) write_csv(*csv_out, al.mate1.name, al.csv_m1);      // alignment.record_csv (aligner_ksw2.hpp:911-914): mate 1's name as remove_slash_mate left it
            if (al.aligned) ++aligned;
        }
        return aligned;
    }
    std::string* csv_out = nullptr;          // -c: where align_batch puts the pairs' lines (include/common/csv.hpp:55-67)

    // st_align's paired loop (align_reads_dispatcher.hpp:356-389): batches of b_size pairs; learn until the model is complete (or the input
    // ends), align the batches read so far, then the rest
    size_t align_all(const std::vector<read_t>& m1, const std::vector<read_t>& m2, size_t b_size, std::string& out) {
        size_t at = 0;
        std::vector<read_t> l1, l2;
        while (at < m1.size()) {
            const size_t e = std::min(m1.size(), at + b_size);
            std::vector<read_t> b1(m1.begin() + at, m1.begin() + e), b2(m2.begin() + at, m2.begin() + e);
            at = e;
            l1.insert(l1.end(), b1.begin(), b1.end()); l2.insert(l2.end(), b2.begin(), b2.end());
            if (learn_fragment_model(b1, b2)) break;
        }
        size_t aligned = align_batch(l1, l2, out);
        while (at < m1.size()) {
            const size_t e = std::min(m1.size(), at + b_size);
            std::vector<read_t> b1(m1.begin() + at, m1.begin() + e), b2(m2.begin() + at, m2.begin() + e);
            at = e;
            aligned += align_batch(b1, b2, out);
        }
        return aligned;
    }
};

}  // namespace oracle
