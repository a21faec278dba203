// Readers and writers for the sdsl-lite serialisations the reference's index files are made of (thirdparty/sdsl-lite is an absent
// submodule; layouts per SURVEY.md App. B).  What is pinned: int_vector<0>, bit_vector, sd_vector<> and select_support_mcl<> are
// reproduced BYTE FOR BYTE on the reference's own fixture data/Chr21.10.ldx (tests/test_ref_index_io.py re-serialises all 28
// sd_vectors of that file and compares with the original bytes), so files written here load in sdsl.  Host code only.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace sdslio {

struct Reader {
    const uint8_t* b; size_t n, o = 0; bool ok = true;
    Reader(const uint8_t* b_, size_t n_) : b(b_), n(n_) {}
    bool need(size_t k) { if (o + k > n) { ok = false; return false; } return true; }
    uint64_t u64() { uint64_t v = 0; if (need(8)) { memcpy(&v, b + o, 8); o += 8; } return v; }
    uint16_t u16() { uint16_t v = 0; if (need(2)) { memcpy(&v, b + o, 2); o += 2; } return v; }
    uint8_t u8() { uint8_t v = 0; if (need(1)) { v = b[o]; o += 1; } return v; }
    // the sizes in a file are not trusted: a bit length beyond what is left of the file fails before any word count is formed from it
    // ((bit_len + 63) / 64 wraps for lengths near 2^64)
    const uint64_t* words(uint64_t bit_len) {
        if (o > n || bit_len > 8ull * (uint64_t)(n - o)) { ok = false; return nullptr; }
        const size_t k = (size_t)((bit_len + 63) / 64);
        if (!need(8 * k)) return nullptr;
        const uint64_t* p = (const uint64_t*)(b + o); o += 8 * k; return p;
    }
};
struct Writer {
    std::vector<uint8_t> out;
    void u64(uint64_t v) { const size_t o = out.size(); out.resize(o + 8); memcpy(out.data() + o, &v, 8); }
    void u16(uint16_t v) { const size_t o = out.size(); out.resize(o + 2); memcpy(out.data() + o, &v, 2); }
    void u8(uint8_t v) { out.push_back(v); }
    void raw(const void* p, size_t k) { const size_t o = out.size(); out.resize(o + k); if (k) memcpy(out.data() + o, p, k); }
};

static inline int hi(uint64_t x) { int r = 0; while (x >>= 1) ++r; return r; }      // sdsl::bits::hi (hi(0) == 0)

// ---- int_vector<0>: u64 bit_len; u8 width; words -------------------------------------------------------------------------------
struct IntVector {
    uint8_t width = 64; uint64_t size = 0; std::vector<uint64_t> w;
    uint64_t get(uint64_t i) const {
        if (width == 0) return 0;
        const uint64_t bit = i * width, k = bit >> 6, sh = bit & 63;
        uint64_t v = w[k] >> sh;
        if (sh + width > 64) v |= w[k + 1] << (64 - sh);
        return width == 64 ? v : v & ((1ull << width) - 1);
    }
    void init(uint64_t n, uint8_t wd) { width = wd; size = n; w.assign((size_t)((n * wd + 63) / 64), 0); }
    void set(uint64_t i, uint64_t v) {
        if (width == 0) return;
        const uint64_t bit = i * width, k = bit >> 6, sh = bit & 63;
        w[k] |= v << sh;
        if (sh + width > 64) w[k + 1] |= v >> (64 - sh);
    }
    bool load(Reader& r) {
        const uint64_t bits = r.u64(); width = r.u8();
        if (width > 64) { r.ok = false; return false; }          // (a shift by more than 63 bits is undefined)
        const uint64_t* p = r.words(bits);
        if (!r.ok) return false;
        size = width ? bits / width : 0;
        w.assign(p, p + (bits + 63) / 64);
        return true;
    }
    void save(Writer& o) const { o.u64(size * width); o.u8(width); o.raw(w.data(), ((size * width + 63) / 64) * 8); }
};

// ---- bit_vector: u64 bit_len; words ------------------------------------------------------------------------------------------------
struct BitVector {
    uint64_t size = 0; std::vector<uint64_t> w;
    void init(uint64_t n) { size = n; w.assign((size_t)((n + 63) / 64), 0); }
    bool get(uint64_t i) const { return (w[i >> 6] >> (i & 63)) & 1; }
    void set(uint64_t i) { w[i >> 6] |= 1ull << (i & 63); }
    bool load(Reader& r) { size = r.u64(); const uint64_t* p = r.words(size); if (!r.ok) return false; w.assign(p, p + (size + 63) / 64); return true; }
    void save(Writer& o) const { o.u64(size); o.raw(w.data(), w.size() * 8); }
};

// ---- select_support_mcl<b> over a bit_vector: written as sdsl builds it, skipped on reading -------------------------------------
static inline bool skip_select(Reader& r) {
    const uint64_t arg = r.u64();
    if (!r.ok) return false;
    if (arg == 0) return true;
    IntVector iv; BitVector bv;
    if (!iv.load(r) || !bv.load(r)) return false;
    for (uint64_t k = 0; k < (arg + 4095) >> 12; ++k) if (!iv.load(r)) return false;
    return true;
}
static inline void write_select(Writer& o, const BitVector& v, bool ones) {
    std::vector<uint64_t> pos;
    for (uint64_t i = 0; i < v.size; ++i) if (v.get(i) == ones) pos.push_back(i);
    const uint64_t arg = pos.size();
    o.u64(arg);
    if (!arg) return;
    const uint64_t capacity = ((v.size + 63) >> 6) << 6;
    const int logn = hi(capacity) + 1;
    const uint64_t logn4 = (uint64_t)logn * logn * logn * logn;
    const uint64_t sb = (arg + 4095) >> 12;
    IntVector super; super.init(sb, (uint8_t)logn);
    std::vector<bool> is_long(sb, false); bool any_long = false;
    for (uint64_t k = 0; k < sb; ++k) {
        const uint64_t first = pos[k * 4096], last = pos[std::min<uint64_t>(arg, (k + 1) * 4096) - 1];
        super.set(k, first);
        if (last - first > logn4) { is_long[k] = true; any_long = true; }
    }
    super.save(o);
    BitVector mol;
    if (any_long) { mol.init(sb); for (uint64_t k = 0; k < sb; ++k) if (is_long[k]) mol.set(k); }
    mol.save(o);
    for (uint64_t k = 0; k < sb; ++k) {
        const uint64_t a0 = k * 4096, a1 = std::min<uint64_t>(arg, a0 + 4096);
        IntVector blk;
        if (is_long[k]) {                    // every position of the superblock, absolute
            blk.init(4096, (uint8_t)(hi(pos[a1 - 1]) + 1));
            for (uint64_t a = a0; a < a1; ++a) blk.set(a - a0, pos[a]);
        } else {                             // every 64th position, relative to the superblock's first
            blk.init(64, (uint8_t)(hi(pos[a1 - 1] - pos[a0]) + 1));
            for (uint64_t a = a0; a < a1; a += 64) blk.set((a - a0) >> 6, pos[a] - pos[a0]);
        }
        blk.save(o);
    }
}

// ---- sd_vector<>: u64 size; u8 wl; int_vector low; bit_vector high; select_support_mcl<1>; select_support_mcl<0> ---------------
struct SdVector {
    uint64_t size = 0; std::vector<uint64_t> ones;      // positions of the ones, increasing
    bool load(Reader& r) {
        size = r.u64(); const uint8_t wl = r.u8();
        IntVector low; BitVector high;
        if (wl > 63 || !low.load(r) || !high.load(r) || !skip_select(r) || !skip_select(r)) return false;
        if (wl && low.width != wl) return false;
        ones.clear();
        uint64_t zeros = 0, k = 0;
        for (uint64_t i = 0; i < high.size; ++i) {
            if (high.get(i)) { if (k >= low.size && wl) return false; ones.push_back((zeros << wl) | (wl ? low.get(k) : 0)); ++k; }
            else ++zeros;
        }
        for (size_t i = 1; i < ones.size(); ++i) if (ones[i] <= ones[i - 1]) return false;
        return ones.empty() || ones.back() < size || size == 0;
    }
    void save(Writer& o) const {
        const uint64_t m = ones.size();
        int logm = hi(m) + 1; const int logn = hi(size) + 1;
        if (logm == logn) --logm;
        const uint8_t wl = (uint8_t)(logn - logm);
        IntVector low; low.init(m, wl);
        BitVector high; high.init(m + (1ull << logm));
        for (uint64_t k = 0; k < m; ++k) {
            if (wl) low.set(k, ones[k] & ((1ull << wl) - 1));
            high.set((ones[k] >> wl) + k);
        }
        o.u64(size); o.u8(wl);
        low.save(o); high.save(o);
        write_select(o, high, true); write_select(o, high, false);
    }
};

}  // namespace sdslio
