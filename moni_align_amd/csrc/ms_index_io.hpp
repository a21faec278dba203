// <prefix>.thrbv.full.lcp.ms — the reference's r-index with thresholds and sampled LCP, as moni_lcp::serialize writes it
// (include/aligner/moni_lcp.hpp:178-225; field order of load() at :208-225):
//
//     u64 terminator_position; u64 256 + u64 F[256]                                   (my_serialize of std::vector<ulint>, common.hpp:407-412)
//     ri::rle_string bwt        = u64 n, R, B; sparse_sd_vector runs; sparse_sd_vector runs_per_letter[256]; huff_string run_heads
//     sparse_sd_vector pred; int_vector<> pred_to_run; int_vector<> samples_last
//     sparse_sd_vector thresholds_per_letter[256]                                     (thresholds_ds.hpp:500-525)
//     sparse_sd_vector pred_start; int_vector<> pred_start_to_run; int_vector<> samples_start; int_vector<> slcp
//
// ri::rle_string, ri::sparse_sd_vector, ri::huff_string (r-index) and sdsl::wt_huff (sdsl-lite) are absent submodules: their
// serialisations are restated from SURVEY.md App. B [UPSTREAM-RECALL] — sparse_sd_vector = u64 u; u64 n; sd_vector if u > 0 (a fork
// writes no n); wt_huff = u64 size; u64 sigma; bit_vector; rank_support_v (an int_vector<64>: u64 bit_len + words, no width byte);
// select_support_mcl<1>; <0>; tree = u64 n_nodes; n x {u64 bv_pos; u64 bv_pos_rank; u16 parent; u16 child[2]}; u16 c_to_leaf[256];
// u64 path[256].  sd_vector / int_vector / bit_vector / select_support_mcl are the ones of sdsl_io.hpp, pinned byte for byte on the
// reference's .ldx fixture.  NO real .thrbv.full.lcp.ms exists in the reference tree, so this layout is UNPINNED; the reader therefore
// (1) tries the layout variants named above and takes the one that consumes the file exactly to its last byte, and (2) checks every
// redundancy the file carries (F against the run heads and lengths, `runs` against the per-letter vectors, pred / pred_to_run against
// the sorted samples, thresholds inside their inter-run intervals) and refuses the file on any disagreement rather than align against
// a misread index.  The writer exists for the round trip (tests/test_ms_index_io.py) and so that an index built here can be handed to
// reference-side tooling; it spells the structures the way sdsl builds them as far as that is known.
// What the aligner needs from the file is only: run heads, run lengths, samples_start, samples_last, thresholds, slcp, F.
#pragma once
#include <algorithm>
#include <numeric>
#include <string>
#include <vector>

#include "ref_index_io.hpp"

namespace msio {

using sdslio::BitVector; using sdslio::IntVector; using sdslio::Reader; using sdslio::SdVector; using sdslio::Writer;

struct MsFile {
    uint64_t terminator_position = 0, n = 0, r = 0, B = 2;
    std::vector<uint64_t> F;                      // 256
    std::vector<uint8_t> heads;                   // r
    std::vector<uint64_t> starts;                 // r + 1
    std::vector<uint64_t> ssa, esa, thr, slcp;    // r each (thr: 0 = none; slcp: as many entries as the file holds, r or r + 1)
    bool has_lcp = true;                          // false: <prefix>.thrbv.full.ms (ms_pointers::serialize, moni.hpp:392-409: no samples of the LCP)
};

struct Variant { bool sparse_has_n; int node_bytes; };

static inline int bitsize(uint64_t x) { return x == 0 ? 1 : sdslio::hi(x) + 1; }      // common.hpp bitsize()

// ---- ri::sparse_sd_vector ---------------------------------------------------------------------------------------------------------
static inline bool load_sparse(Reader& r, const Variant& V, SdVector& v) {
    const uint64_t u = r.u64();
    uint64_t n = 0;
    if (V.sparse_has_n) n = r.u64();
    if (!r.ok) return false;
    v = SdVector();
    if (u == 0) return !V.sparse_has_n || n == 0;
    if (!v.load(r) || v.size != u) return false;
    return !V.sparse_has_n || v.ones.size() == n;
}
static inline void save_sparse(Writer& o, const SdVector& v) {
    o.u64(v.size); o.u64(v.ones.size());
    if (v.size) v.save(o);
}

// ---- sdsl::wt_huff<> (wt_pc over a byte alphabet) ----------------------------------------------------------------------------------
struct WtNode { uint64_t bv_pos, bv_pos_rank; uint16_t parent, child[2]; };
static const uint16_t WT_UNDEF = 0xFFFF;

// the sequence a Huffman-shaped wavelet tree holds: every node's bits split its (order-preserving) subsequence between its children
static inline bool load_wt(Reader& r, const Variant& V, std::vector<uint8_t>& seq) {
    const uint64_t size = r.u64(), sigma = r.u64();
    BitVector bv;
    if (!r.ok || sigma > 256 || !bv.load(r)) return false;
    { const uint64_t bits = r.u64(); if (!r.ok || !r.words(bits)) return false; }      // rank_support_v's basic blocks
    if (!sdslio::skip_select(r) || !sdslio::skip_select(r)) return false;
    const uint64_t nn = r.u64();
    if (!r.ok || nn > 1024 || !r.need(nn * V.node_bytes + 256 * 2 + 256 * 8)) return false;
    std::vector<WtNode> nodes(nn);
    for (uint64_t k = 0; k < nn; ++k) {
        const size_t at = r.o;
        WtNode& x = nodes[k];
        x.bv_pos = r.u64(); x.bv_pos_rank = r.u64(); x.parent = r.u16(); x.child[0] = r.u16(); x.child[1] = r.u16();
        r.o = at + V.node_bytes;
    }
    r.o += 256 * 2 + 256 * 8;          // c_to_leaf, path: derivable from the nodes
    if (size > bv.size && !(size > 0 && nn == 1)) return false;      // every element spends at least one bit of the tree (one-symbol sequence: none) - and at most the file's size in elements
    if (size > 8ull * (uint64_t)r.n) return false;
    seq.assign(size, 0);
    if (size == 0) return true;
    if (nn == 0) return false;
    int root = -1;
    for (uint64_t k = 0; k < nn; ++k) if (nodes[k].parent == WT_UNDEF) { if (root >= 0) return false; root = (int)k; }
    if (root < 0) return false;
    // iterative descent with explicit position lists (depth <= sigma)
    struct Item { int node; std::vector<uint32_t> pos32; std::vector<uint64_t> pos64; };
    const bool wide = size > 0xFFFFFFFFull;
    std::vector<Item> stack;
    { Item it; it.node = root; if (wide) { it.pos64.resize(size); std::iota(it.pos64.begin(), it.pos64.end(), 0ull); } else { it.pos32.resize(size); std::iota(it.pos32.begin(), it.pos32.end(), 0u); } stack.push_back(std::move(it)); }
    uint64_t placed = 0; size_t guard = 0;
    while (!stack.empty()) {
        if (++guard > 4096) return false;
        Item it = std::move(stack.back()); stack.pop_back();
        const WtNode& x = nodes[it.node];
        const uint64_t cnt = wide ? it.pos64.size() : it.pos32.size();
        const bool leaf = x.child[0] == WT_UNDEF;
        if (leaf) {
            if (x.bv_pos_rank > 255) return false;
            const uint8_t c = (uint8_t)x.bv_pos_rank;            // a leaf keeps its symbol in bv_pos_rank
            if (wide) for (uint64_t p : it.pos64) seq[p] = c; else for (uint32_t p : it.pos32) seq[p] = c;
            placed += cnt;
            continue;
        }
        if (x.child[0] >= nn || x.child[1] >= nn || x.bv_pos + cnt > bv.size) return false;
        Item a, b; a.node = x.child[0]; b.node = x.child[1];
        for (uint64_t k = 0; k < cnt; ++k) {
            const bool bit = bv.get(x.bv_pos + k);
            if (wide) (bit ? b : a).pos64.push_back(it.pos64[k]); else (bit ? b : a).pos32.push_back(it.pos32[k]);
        }
        if (wide ? !a.pos64.empty() : !a.pos32.empty()) stack.push_back(std::move(a));
        if (wide ? !b.pos64.empty() : !b.pos32.empty()) stack.push_back(std::move(b));
    }
    return placed == size;
}

// A Huffman-shaped wavelet tree of seq in the wt_pc layout (node bits concatenated in node order, nodes in breadth-first order, root
// first).  The code lengths are Huffman's; which of two equal-weight subtrees goes left is this writer's choice, not known to be sdsl's.
static inline void save_wt(Writer& o, const std::vector<uint8_t>& seq) {
    uint64_t cnt[256] = {0};
    for (uint8_t c : seq) cnt[c]++;
    struct T { uint64_t w; int l, r, sym; };
    std::vector<T> t;
    std::vector<int> live;
    for (int c = 0; c < 256; ++c) if (cnt[c]) { t.push_back(T{cnt[c], -1, -1, c}); live.push_back((int)t.size() - 1); }
    const uint64_t sigma = live.size();
    while (live.size() > 1) {
        std::stable_sort(live.begin(), live.end(), [&](int a, int b) { return t[a].w < t[b].w; });
        const int a = live[0], b = live[1];
        t.push_back(T{t[a].w + t[b].w, a, b, -1});
        live.erase(live.begin(), live.begin() + 2); live.push_back((int)t.size() - 1);
    }
    std::vector<WtNode> nodes; std::vector<int> of;       // breadth-first numbering
    std::vector<uint64_t> path(256, 0); std::vector<uint16_t> c_to_leaf(256, WT_UNDEF);
    BitVector bv;
    if (!live.empty()) {
        std::vector<int> order(1, live[0]); std::vector<int> parent(1, -1);
        for (size_t k = 0; k < order.size(); ++k) if (t[order[k]].l >= 0) { order.push_back(t[order[k]].l); parent.push_back((int)k); order.push_back(t[order[k]].r); parent.push_back((int)k); }
        of.assign(t.size(), -1);
        for (size_t k = 0; k < order.size(); ++k) of[order[k]] = (int)k;
        nodes.resize(order.size());
        uint64_t bits = 0;
        for (size_t k = 0; k < order.size(); ++k) {
            const T& x = t[order[k]];
            WtNode& y = nodes[k];
            y.parent = parent[k] < 0 ? WT_UNDEF : (uint16_t)parent[k];
            if (x.l < 0) { y.child[0] = y.child[1] = WT_UNDEF; y.bv_pos = bits; y.bv_pos_rank = (uint64_t)x.sym; c_to_leaf[x.sym] = (uint16_t)k; }
            else { y.child[0] = (uint16_t)of[x.l]; y.child[1] = (uint16_t)of[x.r]; y.bv_pos = bits; y.bv_pos_rank = 0; bits += x.w; }
        }
        // per symbol: the root-to-leaf path, first decision in the lowest bit; length in the top byte
        for (int c = 0; c < 256; ++c) if (c_to_leaf[c] != WT_UNDEF) {
            uint64_t w = 0, l = 0;
            for (int v = c_to_leaf[c]; nodes[v].parent != WT_UNDEF; v = nodes[v].parent) { w <<= 1; if (nodes[nodes[v].parent].child[1] == v) w |= 1; ++l; }
            path[c] = w | (l << 56);
        }
        bv.init(bits);
        std::vector<uint64_t> fill(nodes.size(), 0);
        for (uint8_t c : seq) {
            const uint64_t p = path[c], l = p >> 56;
            int v = 0;
            for (uint64_t d = 0; d < l; ++d) { const bool bit = (p >> d) & 1; if (bit) bv.set(nodes[v].bv_pos + fill[v]); ++fill[v]; v = nodes[v].child[bit]; }
        }
        uint64_t ones = 0, k = 0;       // bv_pos_rank of inner nodes: ones before bv_pos
        std::vector<size_t> inner;
        for (size_t v = 0; v < nodes.size(); ++v) if (nodes[v].child[0] != WT_UNDEF) inner.push_back(v);
        for (size_t v : inner) { while (k < nodes[v].bv_pos) { ones += bv.get(k); ++k; } nodes[v].bv_pos_rank = ones; }
    }
    o.u64(seq.size()); o.u64(sigma);
    bv.save(o);
    {   // rank_support_v: per 512-bit block the ones before it, then the seven 9-bit counts of the ones before each of its later words
        const uint64_t cap = ((bv.size + 63) >> 6) << 6, nb = ((cap >> 9) + 1) << 1;
        std::vector<uint64_t> bb(nb, 0);
        uint64_t sum = 0;
        for (uint64_t blk = 0; blk * 8 < bv.w.size() + 8 && 2 * blk + 1 < nb; ++blk) {
            bb[2 * blk] = sum;
            uint64_t second = 0, in = 0;
            for (int j = 0; j < 8; ++j) {
                if (j) second |= in << (63 - 9 * j);
                const uint64_t wi = blk * 8 + j;
                if (wi < bv.w.size()) in += (uint64_t)__builtin_popcountll(bv.w[wi]);
            }
            bb[2 * blk + 1] = second;
            sum += in;
        }
        o.u64(nb * 64); o.raw(bb.data(), nb * 8);
    }
    sdslio::write_select(o, bv, true); sdslio::write_select(o, bv, false);
    o.u64(nodes.size());
    for (const WtNode& x : nodes) { o.u64(x.bv_pos); o.u64(x.bv_pos_rank); o.u16(x.parent); o.u16(x.child[0]); o.u16(x.child[1]); }
    o.raw(c_to_leaf.data(), 512); o.raw(path.data(), 2048);
}

// ---- the whole file ---------------------------------------------------------------------------------------------------------------------
static inline bool parse_ms(Reader& rd, const Variant& V, MsFile& M, std::string& err, bool with_lcp = true) {
    M = MsFile();
    M.has_lcp = with_lcp;
    M.terminator_position = rd.u64();
    if (rd.u64() != 256 || !rd.need(256 * 8)) { err = "F is not a vector of 256 words"; return false; }
    M.F.resize(256); memcpy(M.F.data(), rd.b + rd.o, 256 * 8); rd.o += 256 * 8;
    M.n = rd.u64(); M.r = rd.u64(); M.B = rd.u64();
    if (!rd.ok || M.n < 2 || M.r < 1 || M.r > M.n || M.B < 1 || M.B > 1024) { err = "rle_string header"; return false; }
    SdVector runs; std::vector<SdVector> per(256);
    if (!load_sparse(rd, V, runs)) { err = "rle_string::runs"; return false; }
    for (int c = 0; c < 256; ++c) if (!load_sparse(rd, V, per[c])) { err = "rle_string::runs_per_letter"; return false; }
    if (!load_wt(rd, V, M.heads) || M.heads.size() != M.r) { err = "rle_string::run_heads (wt_huff)"; return false; }
    SdVector pred, pred_start; IntVector pred_to_run, pred_start_to_run, s_last, s_start, slcp;
    if (!load_sparse(rd, V, pred) || !pred_to_run.load(rd) || !s_last.load(rd)) { err = "pred / pred_to_run / samples_last"; return false; }
    std::vector<SdVector> thr(256);
    for (int c = 0; c < 256; ++c) if (!load_sparse(rd, V, thr[c])) { err = "thresholds_per_letter"; return false; }
    if (!load_sparse(rd, V, pred_start) || !pred_start_to_run.load(rd) || !s_start.load(rd) || (with_lcp && !slcp.load(rd))) { err = "pred_start / samples_start / slcp"; return false; }
    if (!rd.ok || rd.o != rd.n) { err = "bytes left over"; return false; }
    // ---- redundancy checks and the flat form ----
    const uint64_t r = M.r, n = M.n;
    M.starts.assign(r + 1, 0);
    {   // run lengths from the per-letter vectors (a one at the last position of every c-run, in c-only coordinates: ms_rle_string.hpp:79-95)
        std::vector<uint64_t> at(256, 0);
        for (uint64_t k = 0; k < r; ++k) {
            const uint8_t c = M.heads[k];
            if (at[c] >= per[c].ones.size()) { err = "more runs of a letter than runs_per_letter holds"; return false; }
            const uint64_t end = per[c].ones[at[c]], len = end + 1 - (at[c] ? per[c].ones[at[c] - 1] + 1 : 0);
            ++at[c];
            M.starts[k + 1] = M.starts[k] + len;
        }
        for (int c = 0; c < 256; ++c) if (at[c] != per[c].ones.size()) { err = "runs_per_letter holds more runs than run_heads"; return false; }
        if (M.starts[r] != n) { err = "run lengths do not sum to n"; return false; }
        uint64_t want = 0;            // runs: a one at the last position of every B-th run
        for (uint64_t k = 0; k < r; ++k) if (k % M.B == M.B - 1) { if (want >= runs.ones.size() || runs.ones[want] != M.starts[k + 1] - 1) { err = "rle_string::runs disagrees with the run lengths"; return false; } ++want; }
        if (want != runs.ones.size()) { err = "rle_string::runs disagrees with the run lengths"; return false; }
        std::vector<uint64_t> cnt(256, 0);           // F as build_F_ makes it (moni.hpp:253-282)
        for (uint64_t k = 0; k < r; ++k) cnt[M.heads[k]] += M.starts[k + 1] - M.starts[k];
        uint64_t acc = 0;
        for (int c = 0; c < 256; ++c) { if (M.F[c] != acc) { err = "F disagrees with the BWT"; return false; } acc += cnt[c]; }
    }
    if (s_start.size != r || s_last.size != r || pred_to_run.size != r || pred_start_to_run.size != r || (with_lcp && slcp.size != r && slcp.size != r + 1)) { err = "sample vectors do not hold one entry per run"; return false; }
    M.ssa.resize(r); M.esa.resize(r); M.slcp.resize(with_lcp ? slcp.size : r, 0);
    for (uint64_t k = 0; k < r; ++k) { M.ssa[k] = s_start.get(k); M.esa[k] = s_last.get(k); }
    for (uint64_t k = 0; with_lcp && k < slcp.size; ++k) M.slcp[k] = slcp.get(k);
    auto check_pred = [&](const SdVector& pv, const IntVector& to_run, const std::vector<uint64_t>& samples) {
        if (pv.ones.size() != r || pv.size != n) return false;
        for (uint64_t k = 0; k < r; ++k) { const uint64_t run = to_run.get(k); if (run >= r || samples[run] != pv.ones[k]) return false; }
        return true;
    };
    // moni_lcp builds pred over the .ssa samples and pred_start over the .esa samples (moni_lcp.hpp:99-100)
    if (!check_pred(pred, pred_to_run, M.ssa) || !check_pred(pred_start, pred_start_to_run, M.esa)) { err = "pred / pred_to_run are not the sorted samples"; return false; }
    {   // thresholds: per letter the nonzero ones in run order (thresholds_ds.hpp:413-430); each belongs to the c-run it precedes
        M.thr.assign(r, 0);
        std::vector<std::vector<uint64_t>> runs_of(256);
        for (uint64_t k = 0; k < r; ++k) runs_of[M.heads[k]].push_back(k);
        for (int c = 0; c < 256; ++c) {
            const auto& rc = runs_of[c];
            if (!thr[c].ones.empty() && thr[c].size != n) { err = "threshold universe"; return false; }
            size_t k = 0;
            for (uint64_t v : thr[c].ones) {
                while (k < rc.size() && M.starts[rc[k]] < v) ++k;
                if (k == 0 || k >= rc.size() || M.thr[rc[k]] != 0 || v < M.starts[rc[k - 1] + 1]) { err = "a threshold outside the interval between two runs of its letter"; return false; }
                M.thr[rc[k]] = v;
            }
        }
    }
    return true;
}

static inline int load_ms(const char* path, MsFile& M, std::string& err) {
    std::vector<uint8_t> buf;
    if (!refio::read_file(path, buf) || buf.size() < 8 * 260) { err = "cannot read the file"; return MONI_EIO; }
    const Variant variants[4] = {{true, 22}, {false, 22}, {true, 24}, {false, 24}};
    std::string first;
    for (int with_lcp = 1; with_lcp >= 0; --with_lcp)          // moni_lcp::serialize, then ms_pointers::serialize (the same file without the LCP samples)
        for (const Variant& V : variants) {
            Reader rd(buf.data(), buf.size());
            std::string e;
            if (parse_ms(rd, V, M, e, with_lcp != 0)) { err.clear(); return MONI_OK; }
            if (first.empty()) first = e;
        }
    err = first;
    return MONI_EIO;
}

// header only: n, r and the number of slcp entries are not all in the header; n and r are
static inline int info_ms(const char* path, uint64_t& n, uint64_t& r) {
    FILE* f = fopen(path, "rb");
    if (!f) return MONI_EIO;
    uint64_t h[2 + 256 + 3];
    const bool ok = fread(h, 8, 2 + 256 + 3, f) == 2 + 256 + 3;
    fclose(f);
    if (!ok || h[1] != 256) return MONI_EIO;
    n = h[258]; r = h[259];
    return (n >= 2 && r >= 1 && r <= n) ? MONI_OK : MONI_EIO;
}

static inline SdVector sd_of(std::vector<uint64_t> ones, uint64_t size) { SdVector v; v.size = size; v.ones = std::move(ones); return v; }
static inline void save_int_vector(Writer& o, const std::vector<uint64_t>& v, int width) {
    IntVector iv; iv.init(v.size(), (uint8_t)width);
    for (size_t k = 0; k < v.size(); ++k) iv.set(k, v[k]);
    iv.save(o);
}

static inline int save_ms(const char* path, const moni_flat_index_t& f, uint64_t slcp_len) {
    const uint64_t n = f.n, r = f.r, B = 2;
    Writer o;
    uint64_t term = 0;
    for (uint64_t k = 0; k < r; ++k) if (f.heads[k] <= 1) { term = f.starts[k]; break; }
    o.u64(term); o.u64(256); o.raw(f.F, 256 * 8);
    o.u64(n); o.u64(r); o.u64(B);
    {
        std::vector<uint64_t> ones;
        for (uint64_t k = 0; k < r; ++k) if (k % B == B - 1) ones.push_back(f.starts[k + 1] - 1);
        save_sparse(o, sd_of(std::move(ones), n));
        std::vector<std::vector<uint64_t>> per(256); std::vector<uint64_t> tot(256, 0);
        for (uint64_t k = 0; k < r; ++k) { const uint8_t c = f.heads[k]; tot[c] += f.starts[k + 1] - f.starts[k]; per[c].push_back(tot[c] - 1); }
        for (int c = 0; c < 256; ++c) save_sparse(o, sd_of(std::move(per[c]), tot[c]));
        save_wt(o, std::vector<uint8_t>(f.heads, f.heads + r));
    }
    const int log_n = bitsize(n), log_r = bitsize(r);
    auto save_pred = [&](const uint64_t* samples) {
        std::vector<std::pair<uint64_t, uint64_t>> s(r);
        for (uint64_t k = 0; k < r; ++k) s[k] = {samples[k], k};
        std::sort(s.begin(), s.end());
        std::vector<uint64_t> ones(r), to_run(r);
        for (uint64_t k = 0; k < r; ++k) { ones[k] = s[k].first; to_run[k] = s[k].second; }
        save_sparse(o, sd_of(std::move(ones), n));
        save_int_vector(o, to_run, log_r);
    };
    save_pred(f.ssa);
    save_int_vector(o, std::vector<uint64_t>(f.esa, f.esa + r), log_n);
    {
        std::vector<std::vector<uint64_t>> per(256); std::vector<bool> seen(256, false);
        for (uint64_t k = 0; k < r; ++k) { seen[f.heads[k]] = true; if (f.thr[k]) per[f.heads[k]].push_back(f.thr[k]); }
        for (int c = 0; c < 256; ++c) save_sparse(o, sd_of(std::move(per[c]), seen[c] ? n : 0));
    }
    save_pred(f.esa);
    save_int_vector(o, std::vector<uint64_t>(f.ssa, f.ssa + r), log_n);
    if (f.slcp) {          // (absent: the .thrbv.full.ms form)
        std::vector<uint64_t> s(f.slcp, f.slcp + r);
        while (s.size() < slcp_len) s.push_back(0);
        uint64_t mx = 0; for (uint64_t v : s) mx = std::max(mx, v);
        save_int_vector(o, s, sdslio::hi(mx) + 1);          // sdsl::util::bit_compress
    }
    return refio::write_file(path, o.out) ? MONI_OK : MONI_EIO;
}

}  // namespace msio
