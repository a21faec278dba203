// ORACLE — TEST INFRASTRUCTURE ONLY (see flat_index.hpp header).  PARITY UNPINNED.
//
// seed.hpp: matching-statistics pointers, MEMs and occurrence enumeration, restated from
//   include/ms/moni.hpp:568-624            ms_pointers::_query (thr_bv specialisation)
//   include/aligner/seed_finder.hpp:126-343 find_mems / populate_seed(s) / find_MEM_above/below
//   include/aligner/seed_finder.hpp:377-393 get_prev/next_occ_with_lcp (moni_lcp specialisation)
//   include/aligner/mems.hpp:26-60          mem_t
#pragma once
#include "flat_index.hpp"

namespace oracle {

#define MATE_1 0
#define MATE_2 1
#define MATE_F 0
#define MATE_RC 2

struct mem_t {                       // mems.hpp:31-60
    size_t pos = 0, len = 0, idx = 0, mate = 0, rpos = 0;
    std::vector<size_t> occs;
    size_t total_occ = 0, num_filtered = 0;
    std::map<std::string, size_t> count_dict;
    mem_t(size_t p, size_t l, size_t i, size_t m, size_t r) : pos(p), len(l), idx(i), mate(m), rpos(r) {}
};

struct ms_counters {                 // algorithmic-byte accounting: S, J, P, C
    uint64_t lf_steps = 0, jumps = 0, phi_steps = 0, text_cmp = 0;
};

// moni.hpp:568-624
inline std::vector<size_t> ms_query(const FlatIndex& ix, const char* pattern, const size_t m, ms_counters* cnt = nullptr) {
    std::vector<size_t> ms_pointers(m);
    auto pos = ix.bwt_size() - 1;
    auto sample = ix.get_last_run_sample();
    for (size_t i = 0; i < m; ++i) {
        uint8_t c = (uint8_t)pattern[m - i - 1];
        const auto n_c = ix.number_of_letter(c);
        if (cnt) cnt->lf_steps++;
        if (n_c == 0) {
            sample = 0;
            pos = ix.LF(pos, c);
        } else if (pos < ix.bwt_size() && ix.bwt_at(pos) == c) {
            sample--;
            pos = ix.LF(pos, c);
        } else {
            if (cnt) cnt->jumps++;
            ulint run_of_pos = ix.run_of_position(pos);
            auto rnk_c = ix.run_and_head_rank(run_of_pos, c);
            size_t thr_c = ix.thresholds_rank(pos + 1, c);
            if (rnk_c.first > thr_c) {   // jump up
                size_t run_of_j = ix.run_head_select(rnk_c.first, c);
                sample = ix.samples_last[run_of_j];
                pos = ix.F[c] + rnk_c.second - 1;
            } else {                     // jump down
                size_t run_of_j = ix.run_head_select(rnk_c.first + 1, c);
                sample = ix.samples_start[run_of_j];
                pos = ix.F[c] + rnk_c.second;
            }
        }
        ms_pointers[m - i - 1] = sample;
    }
    return ms_pointers;
}

struct seed_finder {
    const FlatIndex& ix;
    size_t n;                 // ra.getLen()  (seed_finder.hpp:99)
    bool filter_seeds = true;
    size_t n_seeds_thr = 5000;
    size_t min_len = 0;
    ms_counters cnt;

    seed_finder(const FlatIndex& ix_, size_t min_len_, bool filter_seeds_, size_t n_seeds_thr_)
        : ix(ix_), n(ix_.n_text), filter_seeds(filter_seeds_), n_seeds_thr(n_seeds_thr_), min_len(min_len_) {}

    // seed_finder.hpp:126-166
    void find_mems(const char* seq, size_t seq_l, std::vector<mem_t>& mems, size_t r_offset = 0, size_t mate = 0) {
        auto pointers = ms_query(ix, seq, seq_l, &cnt);
        size_t l = 0, pl = 0, n_Ns = 0;
        size_t prev_pos_plus_one = n + 1;
        for (size_t i = 0; i < pointers.size(); ++i) {
            size_t pos = pointers[i];
            // cnt.text_cmp counts every ra.charAt() the reference's loop condition evaluates (matches and the final mismatch)
            while (pos != prev_pos_plus_one && (i + l) < seq_l && (pos + l) < n && (++cnt.text_cmp, (uint8_t)seq[i + l] == ix.text[pos + l])) {
                if (seq[i + l] == 'N') n_Ns++; else n_Ns = 0;
                ++l;
            }
            if (l >= pl and n_Ns < l and l >= min_len) {
                size_t r = r_offset + (i + l - 1);
                mems.push_back(mem_t(pointers[i], l, i, mate, r));
            }
            pl = l;
            l = (l == 0 ? 0 : (l - 1));
            prev_pos_plus_one = pos + 1;
        }
    }

    // lceToRBounded (ShapedSlp, an absent submodule; call sites seed_finder.hpp:354,367): the bytes the text suffixes at a and b share, up to len
    size_t lce_bounded(size_t a, size_t b, size_t len) const {
        size_t l = 0;
        while (l < len && ix.text[a + l] == ix.text[b + l]) ++l;
        return l;
    }
    // seed_finder.hpp:377-393 (moni_lcp<>: the sampled LCP); :346-370 (ms_pointers<>, `-n`: Phi / Phi_inv and a bounded LCE on the text when both
    // suffixes are at least len long)
    std::pair<size_t, size_t> get_next_occ_with_lcp(size_t curr, size_t len) {
        if (curr == ix.get_last_run_sample()) return {ix.get_first_run_sample(), 0};
        cnt.phi_steps++;
        if (!ix.no_lcp) return ix.Phi_inv_lcp(curr);
        const size_t next = ix.Phi_inv_lcp(curr).first;          // ms.Phi_inv(curr): the same predecessor query without the LCP
        size_t lcp = 0;
        if ((n - curr) >= len and (n - next) >= len) lcp = lce_bounded(curr, next, len);
        return {next, lcp};
    }
    std::pair<size_t, size_t> get_prev_occ_with_lcp(size_t curr, size_t len) {
        if (curr == ix.get_first_run_sample()) return {ix.get_last_run_sample(), 0};
        cnt.phi_steps++;
        if (!ix.no_lcp) return ix.Phi_lcp(curr);
        const size_t prev = ix.Phi_lcp(curr).first;
        size_t lcp = 0;
        if ((n - curr) >= len and (n - prev) >= len) lcp = lce_bounded(curr, prev, len);
        return {prev, lcp};
    }

    // seed_finder.hpp:331-343
    size_t populate_dict(size_t pos, std::map<std::string, size_t>& count_dict) {
        std::string ref = ix.name_of(pos);
        auto it = count_dict.find(ref);
        if (it != count_dict.end()) count_dict[ref]++;
        else count_dict[ref] = 1;
        return count_dict[ref];
    }

    // seed_finder.hpp:169-202
    bool find_MEM_above(size_t curr, size_t len, std::vector<size_t>& occs, std::map<std::string, size_t>& count_dict,
                        size_t& total_occ, size_t& num_filtered) {
        auto pl = get_prev_occ_with_lcp(curr, len);
        size_t prev = pl.first, lcp = pl.second;
        while (lcp >= len) {
            size_t ref_count = populate_dict(prev, count_dict);
            occs.push_back(prev);
            total_occ++;
            if (filter_seeds) {
                if (ref_count > n_seeds_thr) { occs.pop_back(); num_filtered++; }
            }
            std::tie(prev, lcp) = get_prev_occ_with_lcp(prev, len);
        }
        return true;
    }
    // seed_finder.hpp:206-239
    bool find_MEM_below(size_t curr, size_t len, std::vector<size_t>& occs, std::map<std::string, size_t>& count_dict,
                        size_t& total_occ, size_t& num_filtered) {
        auto nl = get_next_occ_with_lcp(curr, len);
        size_t next = nl.first, lcp = nl.second;
        while (lcp >= len) {
            size_t ref_count = populate_dict(next, count_dict);
            occs.push_back(next);
            total_occ++;
            if (filter_seeds) {
                if (ref_count > n_seeds_thr) { occs.pop_back(); num_filtered++; }
            }
            std::tie(next, lcp) = get_next_occ_with_lcp(next, len);
        }
        return true;
    }
    // seed_finder.hpp:244-254
    bool find_MEM_occs(mem_t& mem) {
        populate_dict(mem.pos, mem.count_dict);
        mem.occs.push_back(mem.pos);
        mem.total_occ++;
        if (!find_MEM_above(mem.pos, mem.len, mem.occs, mem.count_dict, mem.total_occ, mem.num_filtered)) return false;
        if (!find_MEM_below(mem.pos, mem.len, mem.occs, mem.count_dict, mem.total_occ, mem.num_filtered)) return false;
        return true;
    }

    // seed_finder.hpp:258-308.  NB the reference holds `mem_t &mem` across push_back on the same
    // vector; every field it needs afterwards was copied to locals first, so indexing is equivalent.
    bool populate_seed(size_t j, std::vector<mem_t>& mems, bool report_mems = false) {
        size_t l = mems[j].len, i = mems[j].idx, mate = mems[j].mate, pos = mems[j].pos, r = mems[j].rpos;
        {
            mem_t& mem = mems[j];
            populate_dict(mem.pos, mem.count_dict);
            mem.occs.push_back(mem.pos);
            mem.total_occ++;
        }
        find_MEM_above(mems[j].pos, mems[j].len, mems[j].occs, mems[j].count_dict, mems[j].total_occ, mems[j].num_filtered);
        size_t upper_suffix = mems[j].occs.back();
        find_MEM_below(mems[j].pos, mems[j].len, mems[j].occs, mems[j].count_dict, mems[j].total_occ, mems[j].num_filtered);
        size_t lower_suffix = mems[j].occs.back();

        if (l >= (min_len << 1) && !(report_mems)) {
            size_t ll = l >> 1;
            size_t rl = r - l + ll;
            mems.push_back(mem_t(upper_suffix, ll, i, mate, rl));
            {
                mem_t& mem = mems.back();
                populate_dict(mem.pos, mem.count_dict);
                mem.occs.push_back(upper_suffix);
                mem.total_occ++;
                if ((not find_MEM_above(upper_suffix, mem.len, mem.occs, mem.count_dict, mem.total_occ, mem.num_filtered)) or
                    (not find_MEM_below(lower_suffix, mem.len, mem.occs, mem.count_dict, mem.total_occ, mem.num_filtered))) {
                    mems.pop_back();
                    return false;
                }
            }
            size_t lr = l - ll;
            size_t rr = r;
            mems.push_back(mem_t(pos + ll, lr, i + ll, mate, rr));
            if ((not find_MEM_occs(mems.back()))) {
                mems.pop_back();
                return false;
            }
        }
        return true;
    }

    // seed_finder.hpp:311-318
    void populate_seeds(std::vector<mem_t>& mems, bool report_mems = false) {
        size_t n_MEMs = mems.size();
        for (size_t j = 0; j < n_MEMs; ++j) populate_seed(j, mems, report_mems);
    }
};

}  // namespace oracle
