// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
// PARITY UNPINNED: the reference cannot be compiled here (all 15 third-party submodules are
// empty: sdsl-lite, r-index, ksw2, ShapedSlp, ...), and its own tests hold
// no golden vectors for this path except data/Chr21.10.ldx (+ lifts), which pins only the
// .ldx/sd_vector layout and lift(pos).  Everything else is a restatement of the reference
// source text plus the published algorithms of the absent dependencies (r-index, ksw2).
//
// flat_index.hpp: the semantic content of <prefix>.thrbv.full.lcp.ms / .plain.slp / .ldx as flat
// arrays, with the rank/select primitives the reference reaches through ri::rle_string,
// ri::sparse_sd_vector and ri::huff_string (absent; semantics per SURVEY.md App. B) restated as
// binary searches over those arrays.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace oracle {

typedef uint64_t ulint;

// lift::Lift of levioSAM (absent submodule; SURVEY.md App. B [UPSTREAM-RECALL]): three sd_vectors over the columns of the
// haplotype-vs-reference alignment: ins[x] = the column is an insertion (no reference base), del[x] = the column is a
// deletion (no haplotype base), snp (not needed by this path).  Kept as the sorted positions of the ones; rank/select are
// the sd_vector operations spelled naively.  Pinned by the reference's fixture data/Chr21.10.ldx + data/lifts/*.lft through
// lift_pos only (tests/test_ldx_fixture.py restating test/src/lifting_test.cpp:57-141); lift_cigar is UNPINNED.
struct Lift {
    ulint second = 0;                 // liftidx::lifts[i].second: start of the target contig in the concatenation
    ulint len = 0;                    // number of alignment columns
    std::vector<ulint> ins, del;      // positions of the ones
    static ulint rank1(const std::vector<ulint>& v, ulint i) { return (ulint)(std::lower_bound(v.begin(), v.end(), i) - v.begin()); }
    static bool at(const std::vector<ulint>& v, ulint i) { return std::binary_search(v.begin(), v.end(), i); }
    static ulint select0(const std::vector<ulint>& v, ulint k) {     // position of the k-th zero, k >= 1
        ulint lo = 0, hi = v.size();                                  // j = number of ones before it: zeros before v[j] is v[j] - j
        while (lo < hi) { ulint mid = (lo + hi) >> 1; if (v[mid] - mid < k) lo = mid + 1; else hi = mid; }
        return k - 1 + lo;
    }
    ulint lift_pos(ulint p) const { return del_select0_then_ins_rank0(p); }
    ulint del_select0_then_ins_rank0(ulint p) const { const ulint x = select0(del, p + 1); return x - rank1(ins, x); }   // ins_rs0(del_sls0(p + 1))
    // lift::Lift::lift_cigar / cigar_s2_to_s1 (levioSAM, [UPSTREAM-RECALL]): the CIGAR is expanded to unit operations and
    // walked together with the alignment columns from x = del_sls0(pos + 1): a deleted column emits D and consumes nothing
    // of the read's CIGAR; I stays I; M becomes I on an inserted column, M otherwise; D is dropped on an inserted column.
    std::vector<uint32_t> lift_cigar(const uint32_t* cigar, size_t n_cigar, ulint pos) const {
        std::vector<uint32_t> ops, out_ops, out;
        for (size_t i = 0; i < n_cigar; ++i) for (uint32_t j = 0; j < (cigar[i] >> 4); ++j) ops.push_back(cigar[i] & 0xf);
        ulint x = select0(del, pos + 1);
        size_t y = 0;
        while (y < ops.size()) {
            const uint32_t cop = ops[y];
            if (at(del, x)) { out_ops.push_back(2); ++x; }
            else if (cop == 1) { out_ops.push_back(1); ++y; }
            else if (cop == 0 || cop == 7 || cop == 8) { out_ops.push_back(at(ins, x) ? 1u : 0u); ++x; ++y; }
            else if (cop == 2 || cop == 3) { if (!at(ins, x)) out_ops.push_back(cop); ++x; ++y; }
            else ++y;
        }
        for (size_t i = 0; i < out_ops.size(); ++i) {
            if (!out.empty() && (out.back() & 0xf) == out_ops[i]) out.back() += 1u << 4;
            else out.push_back((1u << 4) | out_ops[i]);
        }
        return out;
    }
};

struct FlatIndex {
    ulint n = 0;        // bwt.size()  (text length + terminator)
    ulint r = 0;        // number of BWT runs
    ulint w = 0;        // separator width
    ulint n_text = 0;   // ra.getLen()
    std::vector<ulint> F;                 // moni.hpp:253-282
    std::vector<uint8_t> heads;           // run heads, bytes <= 1 stored as TERMINATOR(1)
    std::vector<ulint> starts;            // r+1, starts[r] = n
    std::vector<ulint> samples_start;     // moni.hpp:127  value = SA-1 mod n
    std::vector<ulint> samples_last;      // moni.hpp:128
    std::vector<ulint> thr;               // per run, 0 = none (thresholds_ds.hpp:413-426)
    std::vector<ulint> slcp;              // moni_lcp.hpp:117-145
    bool no_lcp = false;                  // the `-n` form: ms_pointers<> instead of moni_lcp<> (align_full_ksw2.cpp:414-419): no LCP samples are consulted
    std::vector<uint8_t> text;            // ra
    std::vector<ulint> seq_starts;        // seqidx onsets (k+1)
    std::vector<std::string> names;
    std::vector<Lift> lifts;              // liftidx::lifts, one per sequence; empty = FASTA-built index (null lifts)

    // derived, mirroring the reference's members
    std::vector<ulint> n_letter;                       // runs_per_letter[c].size()
    std::vector<std::vector<ulint>> crun;              // per letter: global run index of each c-run (run_heads.select)
    std::vector<std::vector<ulint>> crun_cum;          // per letter: runs_per_letter[c].select(j)+1
    std::vector<std::vector<ulint>> thr_per_letter;    // thresholds_per_letter[c] onsets (thresholds_ds.hpp:421-430)
    std::vector<ulint> pred, pred_to_run;              // moni.hpp:186-251 on .ssa
    std::vector<ulint> pred_start, pred_start_to_run;  // moni.hpp:125 on .esa

    void finalize() {
        n_text = n - 1;
        n_letter.assign(256, 0);
        crun.assign(256, {});
        crun_cum.assign(256, {});
        thr_per_letter.assign(256, {});
        for (ulint k = 0; k < r; ++k) {
            uint8_t c = heads[k];
            ulint len = starts[k + 1] - starts[k];
            n_letter[c] += len;
            crun[c].push_back(k);
            crun_cum[c].push_back(n_letter[c]);
            if (thr[k] > 0) thr_per_letter[c].push_back(thr[k]);
        }
        build_phi(samples_start, pred, pred_to_run);
        build_phi(samples_last, pred_start, pred_start_to_run);
    }

    // moni.hpp:186-251
    void build_phi(const std::vector<ulint>& smp, std::vector<ulint>& pv, std::vector<ulint>& pv_to_run) {
        std::vector<std::pair<ulint, ulint>> samples(r);
        for (ulint i = 0; i < r; ++i) samples[i] = std::make_pair(smp[i], i);
        std::sort(samples.begin(), samples.end());
        pv.resize(r);
        pv_to_run.resize(r);
        for (ulint i = 0; i < r; ++i) { pv[i] = samples[i].first; pv_to_run[i] = samples[i].second; }
    }

    // ---- ri::rle_string / ms_rle_string primitives -------------------------------------
    ulint bwt_size() const { return n; }
    ulint number_of_letter(uint8_t c) const { return n_letter[c]; }          // ms_rle_string.hpp:121-124

    // rle_string::run_of_position: for i == n the scan runs off the end and returns R.
    ulint run_of_position(ulint i) const {
        return (ulint)(std::upper_bound(starts.begin(), starts.begin() + r, i) - starts.begin()) - 1
               + (i >= n ? 1 : 0);
    }
    uint8_t bwt_at(ulint i) const { return heads[run_of_position(i)]; }       // rle_string::operator[]

    // run_heads.rank(i, c): number of c among run heads [0, i)
    ulint run_heads_rank(ulint i, uint8_t c) const {
        const auto& v = crun[c];
        return (ulint)(std::lower_bound(v.begin(), v.end(), i) - v.begin());
    }
    // ms_rle_string.hpp:152-160
    std::pair<ulint, ulint> run_and_head_rank(ulint i, uint8_t c) const {
        const ulint j = run_heads_rank(i, c);
        if (j < 1) return std::make_pair(j, j);
        const ulint k = crun_cum[c][j - 1];   // runs_per_letter[c].select(j-1) + 1
        return std::make_pair(j, k);
    }
    // ms_rle_string.hpp:163-167  (i is 1-based)
    ulint run_head_select(ulint i, uint8_t c) const { return crun[c][i - 1]; }

    // rle_string::rank(i, c): number of c in BWT[0, i)
    ulint rank(ulint i, uint8_t c) const {
        ulint run = run_of_position(i);
        ulint rk = run_heads_rank(run, c);
        ulint tail = (run < r && heads[run] == c) ? (i - starts[run]) : 0;
        return (rk == 0 ? 0 : crun_cum[c][rk - 1]) + tail;
    }
    // moni.hpp:319-329
    ulint LF(ulint i, uint8_t c) const { return F[c] + rank(i, c); }

    // thr_bv::rank (thresholds_ds.hpp:494-497): number of thresholds of letter c at positions < i
    ulint thresholds_rank(ulint i, uint8_t c) const {
        const auto& v = thr_per_letter[c];
        return (ulint)(std::lower_bound(v.begin(), v.end(), i) - v.begin());
    }

    ulint get_last_run_sample() const { return (samples_last[r - 1] + 1) % n; }   // r_index (moni.hpp:412-415)
    ulint get_first_run_sample() const { return (samples_start[0] + 1) % n; }     // moni.hpp:331-333

    // sparse_sd_vector::predecessor_rank_circular
    static ulint pred_rank_circular(const std::vector<ulint>& pv, ulint i) {
        ulint rk = (ulint)(std::lower_bound(pv.begin(), pv.end(), i) - pv.begin());   // rank(i) = #1 in [0,i)
        return rk == 0 ? (ulint)pv.size() - 1 : rk - 1;
    }

    // moni_lcp.hpp:253-272
    std::pair<ulint, ulint> Phi_lcp(ulint i) const {
        ulint jr = pred_rank_circular(pred, i);
        ulint j = pred[jr];
        ulint delta = j < i ? i - j : i + 1;
        ulint idx = pred_to_run[jr] - 1;
        if (idx >= r) { fprintf(stderr, "oracle: Phi_lcp on the first run (undefined in the reference)\n"); abort(); }
        ulint prev_sample = samples_last[idx];
        ulint lcp = slcp[idx + 1];
        return std::make_pair((prev_sample + delta) % n, lcp - delta + 1);
    }
    // moni_lcp.hpp:230-248
    std::pair<ulint, ulint> Phi_inv_lcp(ulint i) const {
        ulint jr = pred_rank_circular(pred_start, i);
        ulint j = pred_start[jr];
        ulint delta = j < i ? i - j : i + 1;
        ulint run1 = pred_start_to_run[jr] + 1;
        if (run1 >= r) { fprintf(stderr, "oracle: Phi_inv_lcp on the last run (undefined in the reference)\n"); abort(); }
        ulint prev_sample = samples_start[run1];
        ulint lcp = slcp[run1];
        return std::make_pair((prev_sample + delta) % n, lcp - delta + 1);
    }

    // ---- seqidx (seqidx.hpp:126-180) ---------------------------------------------------
    ulint rank1(ulint i) const {   // number of onsets in [0, i)
        return (ulint)(std::lower_bound(seq_starts.begin(), seq_starts.end(), i) - seq_starts.begin());
    }
    ulint select1(ulint k) const { return seq_starts[k - 1]; }   // 1-based
    ulint seq_length(ulint i) const { return select1(i + 2) - select1(i + 1) - w; }
    ulint seq_of(ulint pos) const { return rank1(pos + 1) - 1; }
    const std::string& name_of(ulint pos) const { return names[rank1(pos + 1) - 1]; }
    std::pair<ulint, ulint> index(ulint pos) const {           // (sequence id, offset)
        ulint rk = rank1(pos + 1);
        return std::make_pair(rk - 1, pos - select1(rk));
    }
    bool valid(ulint pos, ulint len) const { return pos + len <= select1(rank1(pos + 1) + 1); }
    // liftidx.hpp:89-95.  A FASTA-built index (no lifts stored here) carries null lifts over text coordinates: identity.
    ulint lift(ulint pos) const {
        if (lifts.empty()) return pos;
        const ulint rk = rank1(pos + 1);
        const ulint start = pos - select1(rk);
        const Lift& L = lifts[rk - 1];
        return L.second + L.lift_pos(start);
    }
    // liftidx.hpp:159-164 (bam1_t carries the CIGAR and core.pos = offset in its sequence)
    std::vector<uint32_t> lift_cigar(const uint32_t* cigar, size_t n_cigar, ulint pos) const {
        if (lifts.empty()) return Lift().lift_cigar(cigar, n_cigar, 0);      // a null lift still goes through levioSAM's walk: zero-length
        const ulint rk = rank1(pos + 1);                                     // operations disappear and equal neighbours merge
        return lifts[rk - 1].lift_cigar(cigar, n_cigar, pos - select1(rk));
    }

    bool load(const char* path);
};

inline bool FlatIndex::load(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char magic[8];
    uint64_t hdr[6];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "MONIFLT2", 8) != 0) { fclose(f); return false; }
    if (fread(hdr, 8, 6, f) != 6) { fclose(f); return false; }
    n = hdr[0]; r = hdr[1]; w = hdr[2];
    ulint nseq = hdr[3], nblob = hdr[4];
    const bool has_lifts = hdr[5] != 0;
    auto get = [&](void* dst, size_t bytes) {
        if (bytes && fread(dst, 1, bytes, f) != bytes) return false;
        size_t pad = (8 - bytes % 8) % 8;
        char tmp[8];
        if (pad && fread(tmp, 1, pad, f) != pad) return false;
        return true;
    };
    F.resize(256); heads.resize(r); starts.resize(r + 1); samples_start.resize(r); samples_last.resize(r);
    thr.resize(r); slcp.resize(r); text.resize(n - 1); seq_starts.resize(nseq + 1);
    bool ok = get(F.data(), 256 * 8) && get(heads.data(), r) && get(starts.data(), (r + 1) * 8) &&
              get(samples_start.data(), r * 8) && get(samples_last.data(), r * 8) && get(thr.data(), r * 8) &&
              get(slcp.data(), r * 8) && get(text.data(), n - 1) && get(seq_starts.data(), (nseq + 1) * 8);
    std::vector<char> blob(nblob);
    ok = ok && get(blob.data(), nblob);
    lifts.clear();
    if (ok && has_lifts) {
        lifts.resize(nseq);
        for (ulint i = 0; i < nseq && ok; ++i) {
            uint64_t h4[4];
            ok = get(h4, 32);
            if (!ok) break;
            lifts[i].second = h4[0]; lifts[i].len = h4[1];
            lifts[i].ins.resize(h4[2]); lifts[i].del.resize(h4[3]);
            ok = get(lifts[i].ins.data(), h4[2] * 8) && get(lifts[i].del.data(), h4[3] * 8);
        }
    }
    fclose(f);
    if (!ok) return false;
    names.clear();
    size_t p = 0;
    for (ulint i = 0; i < nseq; ++i) {
        uint64_t ln;
        memcpy(&ln, blob.data() + p, 8);
        names.emplace_back(blob.data() + p + 8, blob.data() + p + 8 + ln);
        p += 8 + ln;
    }
    finalize();
    return true;
}

}  // namespace oracle
