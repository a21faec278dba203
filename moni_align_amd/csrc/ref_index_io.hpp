// The reference's on-disk index pieces, read and written without sdsl (host code):
//   <prefix>.ldx   liftidx::serialize (include/aligner/liftidx.hpp:117-143) over seqidx::serialize (include/common/seqidx.hpp:197-238):
//                  u64 u; [u64 w;] sd_vector starts; u64 n_names; n x {u64 len; bytes}; u64 n_lifts; n x {u64 second; lift::Lift}
//                  with lift::Lift = three sd_vectors (ins, del, snp).  Both layouts are taken: the current one with `w`, and the older
//                  one without it that the reference's own fixture data/Chr21.10.ldx has.  Pinned byte for byte by that fixture.
// The r-index file (.thrbv.full.lcp.ms) is in ms_index_io.hpp.
#pragma once
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/moni_hip.h"
#include "sdsl_io.hpp"

namespace refio {

struct LiftSd { uint64_t second = 0; sdslio::SdVector ins, del, snp; };
struct Ldx {
    uint64_t u = 0, w = 0; bool has_w = false;
    sdslio::SdVector starts;
    std::vector<std::string> names;
    std::vector<LiftSd> lifts;
};

static inline bool read_file(const char* path, std::vector<uint8_t>& buf) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return false; }
    buf.resize((size_t)n);
    const bool ok = n == 0 || fread(buf.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}
static inline bool write_file(const char* path, const std::vector<uint8_t>& buf) {
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    const bool ok = buf.empty() || fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    return fclose(f) == 0 && ok;
}

// the body after u [w]; returns false if it does not parse to the last byte
static inline bool parse_ldx_body(sdslio::Reader& r, Ldx& L) {
    if (!L.starts.load(r)) return false;
    const uint64_t nn = r.u64();
    if (!r.ok || nn > r.n) return false;
    L.names.clear();
    for (uint64_t i = 0; i < nn; ++i) {
        const uint64_t ln = r.u64();
        if (!r.need(ln)) return false;
        L.names.emplace_back((const char*)r.b + r.o, (size_t)ln); r.o += ln;
    }
    const uint64_t nl = r.u64();
    if (!r.ok || nl > r.n) return false;
    L.lifts.assign(nl, LiftSd());
    for (uint64_t i = 0; i < nl; ++i) {
        L.lifts[i].second = r.u64();
        if (!L.lifts[i].ins.load(r) || !L.lifts[i].del.load(r) || !L.lifts[i].snp.load(r)) return false;
    }
    return r.ok && r.o == r.n && L.starts.ones.size() == L.names.size() + 1;
}

static inline int load_ldx(const char* path, Ldx& L) {
    std::vector<uint8_t> buf;
    if (!read_file(path, buf) || buf.size() < 16) return MONI_EIO;
    // current layout first (u, w), then the older one (u): exactly one of them consumes the file to its last byte
    for (int with_w = 1; with_w >= 0; --with_w) {
        sdslio::Reader r(buf.data(), buf.size());
        L = Ldx();
        L.u = r.u64(); L.has_w = with_w != 0;
        if (with_w) L.w = r.u64();
        if (with_w && L.w > 4096) continue;
        if (parse_ldx_body(r, L)) return MONI_OK;
    }
    return MONI_EIO;
}
static inline int save_ldx(const char* path, const Ldx& L, bool with_w) {
    sdslio::Writer o;
    o.u64(L.u);
    if (with_w) o.u64(L.w);
    L.starts.save(o);
    o.u64(L.names.size());
    for (const auto& s : L.names) { o.u64(s.size()); o.raw(s.data(), s.size()); }
    o.u64(L.lifts.size());
    for (const auto& l : L.lifts) { o.u64(l.second); l.ins.save(o); l.del.save(o); l.snp.save(o); }
    return write_file(path, o.out) ? MONI_OK : MONI_EIO;
}

// .ldx content as the flat arrays moni_flat_index_t carries (the arrays live in `hold`)
struct LdxFlat {
    std::vector<uint64_t> seq_starts, second, len, ins_off, ins, del_off, del;
    std::string names;          // NUL-separated
    void from(const Ldx& L) {
        seq_starts = L.starts.ones;
        ins_off.assign(1, 0); del_off.assign(1, 0);
        for (const auto& l : L.lifts) {
            second.push_back(l.second); len.push_back(l.ins.size);
            ins.insert(ins.end(), l.ins.ones.begin(), l.ins.ones.end()); ins_off.push_back(ins.size());
            del.insert(del.end(), l.del.ones.begin(), l.del.ones.end()); del_off.push_back(del.size());
        }
        ins.push_back(0); del.push_back(0);
        for (const auto& s : L.names) { names += s; names.push_back('\0'); }
    }
    void fill(moni_flat_index_t& f, uint64_t w) const {
        f.n_seq = seq_starts.size() - 1; f.w = w; f.seq_starts = seq_starts.data(); f.seq_names = names.c_str();
        f.lift_second = second.data(); f.lift_len = len.data(); f.lift_ins_off = ins_off.data(); f.lift_ins = ins.data();
        f.lift_del_off = del_off.data(); f.lift_del = del.data();
    }
};

}  // namespace refio
