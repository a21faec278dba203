// ORACLE — TEST INFRASTRUCTURE ONLY (see flat_index.hpp header).  PARITY UNPINNED.
// C entry points so that tests/ and bench.py's cpu_baseline leg can drive the restatement
// through ctypes.  Not linked into, loaded by, or called from the product library.
#include "flat_index.hpp"
#include "seed.hpp"
#include "ksw2.hpp"
#include "align.hpp"
#include "align_pe.hpp"

#include <atomic>
#include <chrono>
#include <thread>

using namespace oracle;

namespace {

// kpbseq.h:120-137
static unsigned char seq_compl_table[256];
struct InitTables {
    InitTables() {
        for (int i = 0; i < 256; ++i) seq_compl_table[i] = (unsigned char)i;
        seq_compl_table['A'] = 'T'; seq_compl_table['C'] = 'G'; seq_compl_table['G'] = 'C'; seq_compl_table['T'] = 'A';
        seq_compl_table['a'] = 'T'; seq_compl_table['c'] = 'G'; seq_compl_table['g'] = 'C'; seq_compl_table['t'] = 'A';
    }
} init_tables;

struct SeedResult {
    // one row per MEM, in the order of the reference's per-read `mems` vector
    std::vector<uint64_t> mem_read, mem_pos, mem_len, mem_idx, mem_mate, mem_rpos, mem_total_occ, mem_num_filtered,
        mem_occ_off, mem_occ_cnt;
    std::vector<uint64_t> occs;
    std::vector<uint64_t> read_mem_off;   // n_reads + 1
    ms_counters cnt;
};

static void seed_one(const FlatIndex& ix, seed_finder& sf, const char* s, size_t l, std::vector<mem_t>& mems) {
    // aligner_ksw2.hpp:169-176 (rc copy), 333-337
    std::string rc(l, 0);
    for (size_t i = 0; i < l; ++i) rc[i] = (char)seq_compl_table[(unsigned char)s[l - i - 1]];
    sf.find_mems(s, l, mems, 0, MATE_1 | MATE_F);
    sf.find_mems(rc.data(), l, mems, 0, MATE_1 | MATE_RC);
    sf.populate_seeds(mems);
}

}  // namespace

extern "C" {

void* orc_index_load(const char* path) {
    FlatIndex* ix = new FlatIndex();
    if (!ix->load(path)) { delete ix; return nullptr; }
    return ix;
}

// build from host arrays (same content as the MONIFLT2 file); names: n_seq NUL-terminated strings back to back
void* orc_index_create(uint64_t n, uint64_t r, uint64_t w, uint64_t n_seq, const uint64_t* F, const uint8_t* heads,
                       const uint64_t* starts, const uint64_t* ssa, const uint64_t* esa, const uint64_t* thr,
                       const uint64_t* slcp, const uint8_t* text, const uint64_t* seq_starts, const char* names) {
    FlatIndex* ix = new FlatIndex();
    ix->n = n; ix->r = r; ix->w = w;
    ix->F.assign(F, F + 256); ix->heads.assign(heads, heads + r); ix->starts.assign(starts, starts + r + 1);
    ix->samples_start.assign(ssa, ssa + r); ix->samples_last.assign(esa, esa + r); ix->thr.assign(thr, thr + r);
    ix->slcp.assign(slcp, slcp + r); ix->text.assign(text, text + n - 1); ix->seq_starts.assign(seq_starts, seq_starts + n_seq + 1);
    const char* p = names;
    for (uint64_t i = 0; i < n_seq; ++i) { ix->names.emplace_back(p); p += ix->names.back().size() + 1; }
    ix->finalize();
    return ix;
}
// the `-n` form of the aligner (seed_finder<slp_t, ms_pointers<>>, align_full_ksw2.cpp:414-419): no sampled LCP in the occurrence walks
void orc_index_set_no_lcp(void* h, int no_lcp) { ((FlatIndex*)h)->no_lcp = no_lcp != 0; }
// liftidx::lifts as flat arrays (one lift per sequence): second, number of alignment columns, sorted positions of the ones of
// the ins / del bit-vectors (ragged, offsets n_seq + 1)
void orc_index_set_lifts(void* h, uint64_t n_seq, const uint64_t* second, const uint64_t* len, const uint64_t* ins_off, const uint64_t* ins,
                         const uint64_t* del_off, const uint64_t* del) {
    FlatIndex* ix = (FlatIndex*)h;
    ix->lifts.assign(n_seq, Lift());
    for (uint64_t i = 0; i < n_seq; ++i) {
        Lift& L = ix->lifts[i];
        L.second = second[i]; L.len = len[i];
        L.ins.assign(ins + ins_off[i], ins + ins_off[i + 1]);
        L.del.assign(del + del_off[i], del + del_off[i + 1]);
    }
}
uint64_t orc_lift(void* h, uint64_t pos) { return ((FlatIndex*)h)->lift(pos); }
// lifted CIGAR of an alignment that starts at text position pos; returns the number of operations (out has room for cap)
uint64_t orc_lift_cigar(void* h, const uint32_t* cigar, uint64_t n_cigar, uint64_t pos, uint32_t* out, uint64_t cap) {
    const std::vector<uint32_t> v = ((FlatIndex*)h)->lift_cigar(cigar, n_cigar, pos);
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return v.size();
}
void orc_index_free(void* h) { delete (FlatIndex*)h; }
uint64_t orc_index_n(void* h) { return ((FlatIndex*)h)->n; }
uint64_t orc_index_r(void* h) { return ((FlatIndex*)h)->r; }

void orc_ms_query(void* h, const char* p, uint64_t m, uint64_t* out) {
    auto v = ms_query(*(FlatIndex*)h, p, m);
    for (uint64_t i = 0; i < m; ++i) out[i] = v[i];
}

void orc_phi_lcp(void* h, uint64_t i, int inverse, uint64_t* out2) {
    auto pr = inverse ? ((FlatIndex*)h)->Phi_inv_lcp(i) : ((FlatIndex*)h)->Phi_lcp(i);
    out2[0] = pr.first; out2[1] = pr.second;
}

// seeds (MEMs + occurrences) for a ragged batch of reads, T threads over contiguous read ranges
void* orc_seed_batch(void* h, const uint8_t* seqs, const uint64_t* offsets, uint64_t n_reads,
                     uint64_t min_len, int filter_seeds, uint64_t n_seeds_thr, int threads) {
    const FlatIndex& ix = *(FlatIndex*)h;
    if (threads < 1) threads = 1;
    std::vector<SeedResult> part(threads);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) {
        th.emplace_back([&, t]() {
            uint64_t lo = n_reads * t / threads, hi = n_reads * (t + 1) / threads;
            seed_finder sf(ix, min_len, filter_seeds != 0, n_seeds_thr);
            SeedResult& R = part[t];
            for (uint64_t rd = lo; rd < hi; ++rd) {
                std::vector<mem_t> mems;
                seed_one(ix, sf, (const char*)seqs + offsets[rd], offsets[rd + 1] - offsets[rd], mems);
                R.read_mem_off.push_back(R.mem_pos.size());
                for (auto& m : mems) {
                    R.mem_read.push_back(rd); R.mem_pos.push_back(m.pos); R.mem_len.push_back(m.len);
                    R.mem_idx.push_back(m.idx); R.mem_mate.push_back(m.mate); R.mem_rpos.push_back(m.rpos);
                    R.mem_total_occ.push_back(m.total_occ); R.mem_num_filtered.push_back(m.num_filtered);
                    R.mem_occ_off.push_back(R.occs.size()); R.mem_occ_cnt.push_back(m.occs.size());
                    for (auto o : m.occs) R.occs.push_back(o);
                }
            }
            R.cnt = sf.cnt;
        });
    }
    for (auto& x : th) x.join();
    SeedResult* out = new SeedResult();
    for (int t = 0; t < threads; ++t) {
        SeedResult& R = part[t];
        uint64_t mbase = out->mem_pos.size(), obase = out->occs.size();
        for (auto v : R.read_mem_off) out->read_mem_off.push_back(v + mbase);
        for (auto v : R.mem_occ_off) out->mem_occ_off.push_back(v + obase);
#define APP(f) out->f.insert(out->f.end(), R.f.begin(), R.f.end())
        APP(mem_read); APP(mem_pos); APP(mem_len); APP(mem_idx); APP(mem_mate); APP(mem_rpos);
        APP(mem_total_occ); APP(mem_num_filtered); APP(mem_occ_cnt); APP(occs);
#undef APP
        out->cnt.lf_steps += R.cnt.lf_steps; out->cnt.jumps += R.cnt.jumps;
        out->cnt.phi_steps += R.cnt.phi_steps; out->cnt.text_cmp += R.cnt.text_cmp;
    }
    out->read_mem_off.push_back(out->mem_pos.size());
    return out;
}
uint64_t orc_seed_n_mems(void* r) { return ((SeedResult*)r)->mem_pos.size(); }
uint64_t orc_seed_n_occs(void* r) { return ((SeedResult*)r)->occs.size(); }
// fields: 0 read,1 pos,2 len,3 idx,4 mate,5 rpos,6 total_occ,7 num_filtered,8 occ_off,9 occ_cnt,10 occs,11 read_mem_off,12 counters
void orc_seed_get(void* r, int field, uint64_t* out) {
    SeedResult* R = (SeedResult*)r;
    const std::vector<uint64_t>* v = nullptr;
    switch (field) {
        case 0: v = &R->mem_read; break; case 1: v = &R->mem_pos; break; case 2: v = &R->mem_len; break;
        case 3: v = &R->mem_idx; break; case 4: v = &R->mem_mate; break; case 5: v = &R->mem_rpos; break;
        case 6: v = &R->mem_total_occ; break; case 7: v = &R->mem_num_filtered; break;
        case 8: v = &R->mem_occ_off; break; case 9: v = &R->mem_occ_cnt; break; case 10: v = &R->occs; break;
        case 11: v = &R->read_mem_off; break;
        case 12: out[0] = R->cnt.lf_steps; out[1] = R->cnt.jumps; out[2] = R->cnt.phi_steps; out[3] = R->cnt.text_cmp; return;
    }
    if (v && !v->empty()) memcpy(out, v->data(), v->size() * 8);
}
void orc_seed_free(void* r) { delete (SeedResult*)r; }


// Full single-end `moni align` path for a ragged batch of reads -> SAM text (no header unless asked).
// names: ragged bytes with name_off[n_reads+1]; quals: same offsets as seqs or NULL (FASTA reads print '*').
// counters[8]: lf_steps, jumps, phi_steps, text_cmp, dp_cells, dp_calls, ref_bytes, aligned_reads
char* orc_align_batch(void* h, const uint8_t* seqs, const uint64_t* offsets, uint64_t n_reads, const uint8_t* names,
                      const uint64_t* name_off, const uint8_t* quals, int with_header, int threads, uint64_t* out_len,
                      uint64_t* counters) {
    const FlatIndex& ix = *(FlatIndex*)h;
    if (threads < 1) threads = 1;
    std::vector<std::string> parts(threads);
    std::vector<std::vector<uint64_t>> cnt(threads, std::vector<uint64_t>(8, 0));
    std::vector<std::thread> th;
    align_config_t cfg;
    for (int t = 0; t < threads; ++t) {
        th.emplace_back([&, t]() {
            uint64_t lo = n_reads * t / threads, hi = n_reads * (t + 1) / threads;
            aligner A(ix, cfg);
            std::string& out = parts[t];
            for (uint64_t rd = lo; rd < hi; ++rd) {
                read_t r;
                r.name.assign((const char*)names + name_off[rd], (const char*)names + name_off[rd + 1]);
                r.seq.assign((const char*)seqs + offsets[rd], (const char*)seqs + offsets[rd + 1]);
                if (quals) { r.qual.assign((const char*)quals + offsets[rd], (const char*)quals + offsets[rd + 1]); r.has_qual = true; }
                if (A.align_read(r, out)) cnt[t][7]++;
            }
            cnt[t][0] = A.mem_finder.cnt.lf_steps; cnt[t][1] = A.mem_finder.cnt.jumps; cnt[t][2] = A.mem_finder.cnt.phi_steps;
            cnt[t][3] = A.mem_finder.cnt.text_cmp; cnt[t][4] = A.kc.cells; cnt[t][5] = A.kc.calls; cnt[t][6] = A.dpc.ref_bytes;
        });
    }
    for (auto& x : th) x.join();
    std::string all;
    if (with_header) { aligner A(ix, cfg); all = A.sam_header(); }
    for (auto& p : parts) all += p;
    if (counters) for (int i = 0; i < 8; ++i) { counters[i] = 0; for (int t = 0; t < threads; ++t) counters[i] += cnt[t][i]; }
    char* buf = (char*)malloc(all.size() + 1);
    memcpy(buf, all.data(), all.size());
    buf[all.size()] = 0;
    *out_len = all.size();
    return buf;
}
// `-c`: the CSV lines of the batch (include/common/csv.hpp:55-67; no header), one thread
char* orc_align_csv(void* h, const uint8_t* seqs, const uint64_t* offsets, uint64_t n_reads, const uint8_t* names, const uint64_t* name_off, uint64_t* out_len) {
    const FlatIndex& ix = *(FlatIndex*)h;
    align_config_t cfg;
    aligner A(ix, cfg);
    std::string sam, csv;
    for (uint64_t rd = 0; rd < n_reads; ++rd) {
        read_t r;
        r.name.assign((const char*)names + name_off[rd], (const char*)names + name_off[rd + 1]);
        r.seq.assign((const char*)seqs + offsets[rd], (const char*)seqs + offsets[rd + 1]);
        A.align_read(r, sam, &csv);
    }
    char* buf = (char*)malloc(csv.size() + 1);
    memcpy(buf, csv.data(), csv.size());
    buf[csv.size()] = 0;
    *out_len = csv.size();
    return buf;
}
// Paired-end path (align_pe.hpp; find_orphan == 0: the reference with -u), one thread, st_align's batch order: mate k of pair i is
// read i of batch k.  out[0] = aligned pairs, out[1..4] = the learnt insert-size model (count, mean, std dev, complete).
// (bit 4 of find_orphan: -c - the text returned is the pairs' CSV lines, include/common/csv.hpp:55-67, instead of the SAM records)
char* orc_align_pe(void* h, const uint8_t* seqs1, const uint64_t* off1, const uint8_t* seqs2, const uint64_t* off2, uint64_t n_pairs,
                   const uint8_t* names1, const uint64_t* noff1, const uint8_t* names2, const uint64_t* noff2, const uint8_t* quals1,
                   const uint8_t* quals2, uint64_t b_size, int find_orphan, uint64_t* out_len, double* out) {
    const FlatIndex& ix = *(FlatIndex*)h;
    align_config_t cfg;
    const bool want_csv = (find_orphan & 16) != 0;
    find_orphan &= 15;
    if (find_orphan & 2) { cfg.report_mems = true; find_orphan &= 13; }          // bit 1: -m, the MEM records of the pairs instead of their alignments
    pe_config_t pcfg;
    if (find_orphan & 8) { pcfg.secondary_chains = true; find_orphan &= 7; }   // bit 3: -Z
    if (find_orphan & 4) { pcfg.filter_dir = false; find_orphan &= 3; }         // bit 2: --no-filter-dir
    pcfg.find_orphan = (find_orphan & 1) != 0;
    aligner_pe A(ix, cfg, pcfg);
    std::vector<read_t> m1(n_pairs), m2(n_pairs);
    for (uint64_t i = 0; i < n_pairs; ++i) {
        m1[i].name.assign((const char*)names1 + noff1[i], (const char*)names1 + noff1[i + 1]);
        m2[i].name.assign((const char*)names2 + noff2[i], (const char*)names2 + noff2[i + 1]);
        m1[i].seq.assign((const char*)seqs1 + off1[i], (const char*)seqs1 + off1[i + 1]);
        m2[i].seq.assign((const char*)seqs2 + off2[i], (const char*)seqs2 + off2[i + 1]);
        if (quals1) { m1[i].qual.assign((const char*)quals1 + off1[i], (const char*)quals1 + off1[i + 1]); m1[i].has_qual = true; }
        if (quals2) { m2[i].qual.assign((const char*)quals2 + off2[i], (const char*)quals2 + off2[i + 1]); m2[i].has_qual = true; }
    }
    std::string all, csv;
    if (want_csv) A.csv_out = &csv;
    const size_t aligned = A.align_all(m1, m2, b_size ? b_size : 512, all);
    if (want_csv) all.swap(csv);
    if (out) { out[0] = (double)aligned; out[1] = (double)A.ins_count; out[2] = A.ins_mean; out[3] = A.ins_std_dev; out[4] = A.ins_learning_complete ? 1.0 : 0.0; out[5] = (double)A.orphan_pairs; out[6] = (double)A.orphan_recovered; }
    char* buf = (char*)malloc(all.size() + 1);
    memcpy(buf, all.data(), all.size());
    buf[all.size()] = 0;
    *out_len = all.size();
    return buf;
}
// klib's ksw_align (KSW_XSTART) as restated in align_pe.hpp: out = score, te, qe, tb, qb
void orc_ksw_align(const uint8_t* q, int qlen, const uint8_t* t, int tlen, const int8_t* mat, int gapo, int gape, int* out) {
    std::vector<uint8_t> qq(q, q + qlen), tt(t, t + tlen);
    const aligner_pe::kswr_t r = aligner_pe::ksw_align(qlen, qq.data(), tlen, tt.data(), 5, mat, gapo, gape);
    out[0] = r.score; out[1] = r.te; out[2] = r.qe; out[3] = r.tb; out[4] = r.qb;
}
// aligner::align with report_mems (-m): SAM text of the MEM records of a ragged batch
char* orc_report_mems_batch(void* h, const uint8_t* seqs, const uint64_t* offsets, uint64_t n_reads, const uint8_t* names,
                            const uint64_t* name_off, const uint8_t* quals, uint64_t* out_len) {
    const FlatIndex& ix = *(FlatIndex*)h;
    align_config_t cfg;
    cfg.report_mems = true;
    aligner A(ix, cfg);
    std::string all;
    for (uint64_t rd = 0; rd < n_reads; ++rd) {
        read_t r;
        r.name.assign((const char*)names + name_off[rd], (const char*)names + name_off[rd + 1]);
        r.seq.assign((const char*)seqs + offsets[rd], (const char*)seqs + offsets[rd + 1]);
        if (quals) { r.qual.assign((const char*)quals + offsets[rd], (const char*)quals + offsets[rd + 1]); r.has_qual = true; }
        A.align_read(r, all);
    }
    char* buf = (char*)malloc(all.size() + 1);
    memcpy(buf, all.data(), all.size());
    buf[all.size()] = 0;
    *out_len = all.size();
    return buf;
}

// legacy `moni ms` (src/matching_statistics.cpp:242-256): pointers of ms.query and the lengths loop
void orc_ms_lengths(void* h, const char* p, uint64_t m, uint64_t* ptr_out, uint64_t* len_out) {
    const FlatIndex& ix = *(FlatIndex*)h;
    auto pointers = ms_query(ix, p, m);
    const uint64_t n = ix.n_text;
    uint64_t l = 0;
    for (uint64_t i = 0; i < m; ++i) {
        const uint64_t pos = pointers[i];
        while ((i + l) < m && (pos + l) < n && (i < 1 || pos != (pointers[i - 1] + 1)) && (uint8_t)p[i + l] == ix.text[pos + l]) ++l;
        ptr_out[i] = pos; len_out[i] = l;
        l = (l == 0 ? 0 : (l - 1));
    }
}

void orc_free(void* p) { free(p); }

// ksw2 restatement: one problem. out[11] = max,max_q,max_t,mqe,mqe_t,mte,mte_q,score,reach_end,n_cigar,zdropped
void orc_extz(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int8_t m, const int8_t* mat,
              int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, int32_t* out, uint32_t* cigar, int cigar_cap) {
    ksw_extz_t ez;
    memset(&ez, 0, sizeof(ez));
    ksw_extz2_restated(qlen, query, tlen, target, m, mat, q, e, w, zdrop, end_bonus, flag, &ez);
    out[0] = (int32_t)ez.max; out[1] = ez.max_q; out[2] = ez.max_t; out[3] = ez.mqe; out[4] = ez.mqe_t;
    out[5] = ez.mte; out[6] = ez.mte_q; out[7] = ez.score; out[8] = ez.reach_end; out[9] = ez.n_cigar;
    out[10] = ez.zdropped;
    for (int i = 0; i < ez.n_cigar && i < cigar_cap; ++i) cigar[i] = ez.cigar[i];
    free(ez.cigar);
}

}  // extern "C"
