// align_kernel: the whole per-read path after seeding, persistent waves, AK_NL reads in flight per wavefront.  Every lane
// runs one read's state machine (align_core.h: frequency filter, chaining with the libstdc++ sort emulation, chain-selection
// loop, fill_chain, CIGAR stitching) out of its own slot in HBM; whenever reads need DP results the whole wave solves their
// problems one after the other (extz_wave_lds_lite: LDS H/E/F tiles), operands taken in place from the resident reads and
// the index text, identical problems of a read answered from a memo.  Seeds are read where the seeding kernels left them in
// HBM: nothing goes to the host between stages.  Output per read: a fixed record plus CIGAR / alternative-hit / MD text
// entries in bump-allocated pools (pinned host memory); MAPQ and SAM text are host work.
// A read that does not fit the kernel's capacities is marked (status 2) and taken by the host pipeline (align_host.hpp).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "align_core.h"

#define AK_CIG_CAP 4096u                     // CIGAR entries of one round's traceback problems (each takes qlen + tlen + 2)
#define AK_DIRS_CAP (384u * 1024u)           // direction bytes of one CIGAR problem
#define AK_CUR 32                            // statistics / cursor words per launch
#define AK_TXT_CAP 4096                      // bytes of SAM text per read formatted in the kernel (longer: the host formats the record)
#define AK_MD_CAP 1536                       // bytes of MD text per read (longer: the host pipeline takes the read)
#define AK_MEMO 24                           // score-only DP results remembered per read

struct moni_aln_rec_t {                      // one per read
    uint32_t status;                         // 0 not aligned, 1 aligned, 2 overflow (host pipeline)
    uint32_t strand;
    uint64_t ref_pos;
    int32_t score, score2;
    uint32_t n_cigar, n_alt;
    uint64_t cigar_off, alt_off;             // into the pools
    int32_t nm;                              // NM and the MD:Z text of the lifted alignment (write_MD_core, sam.hpp:249-287), computed where the read and the text are resident
    uint32_t md_len;
    uint64_t md_off;                         // into the MD pool (8-byte aligned)
    uint32_t txt_len; int32_t lift_nm;       // the finished SAM line, when the kernel formats text (ak_args_t::fmt.txt_pool); NM of the unlifted alignment (OA tag)
    uint64_t txt_off;                        // into the text pool, in 8-byte words
};
struct moni_alt_t { uint64_t pos; int32_t score; int32_t pad; };

#ifndef AK_NL
#define AK_NL 64                             // reads in flight per wavefront: lanes 0..AK_NL-1 each run one read's state machine
#endif
#ifndef AK_START_MIN
#define AK_START_MIN (AK_NL / 2)             // free lanes take new reads together, once this many are free (or nobody waits for DP):
#endif                                       // seeds -> chains is the long lane-private part, it has to run on many lanes at once

struct ak_slot_t {                           // per read in flight, in HBM
    ac_ws_t ws;
    moni_dp_result_t res[AC_MAX_TASKS];      // results of the round's DP problems
    uint32_t cig[AK_CIG_CAP];
    uint32_t lcig[AC_MAX_CIGAR];             // the final CIGAR lifted to the reference contig (liftidx::lift_cigar)
    uint64_t md_tmp[AK_MD_CAP / 8];
    uint64_t txt_tmp[AK_TXT_CAP / 8];
    uint64_t memo_key[AK_MEMO], memo_toff[AK_MEMO];
    dp_brief_t memo_val[AK_MEMO];           // mqe, mqe_t, score of the remembered problem
    uint32_t memo_n, pad;
};
struct ak_wave_t { uint8_t dirs[AK_DIRS_CAP]; };      // per wavefront: direction bytes of the CIGAR problem being solved

// Two DP problems of one read with the same query segment, flags and target length have the same result when their target
// windows spell the same nt4 string - the usual case when a read's chains lie on haplotypes that agree around the read.
// Wave-wide comparison of the two windows, addressed exactly as extz_wave_lds addresses them.
__device__ __attribute__((noinline)) bool ak_same_target(const dp_launch_t& D, int mode, uint64_t t1, uint64_t t2, int tlen) {
    const uint8_t* __restrict__ text = D.text;
    const uint64_t n_text = D.n_text;
    bool diff = false;
    for (int i = threadIdx.x & 63; i < tlen; i += 64) {
        const uint64_t a1 = (mode & DP_T_REV) ? t1 - (uint64_t)i : t1 + (uint64_t)i;
        const uint64_t a2 = (mode & DP_T_REV) ? t2 - (uint64_t)i : t2 + (uint64_t)i;
        diff |= dp_nt4(a1 < n_text ? text[a1] : 0u) != dp_nt4(a2 < n_text ? text[a2] : 0u);
    }
    return __ballot(diff) == 0ull;
}

struct ak_fmt_t {                            // what the kernel needs to spell a SAM line (sam.hpp:144-188): all device memory
    const uint8_t* rnames; const uint64_t* rname_off;    // read names of the resident batch, ragged
    const uint8_t* quals;                                // qualities, same offsets as the reads, or nullptr ('*')
    const uint8_t* snames; const uint32_t* sname_off;    // sequence names of the index, ragged [n_seq + 1]
    const double* mapq_tab; uint32_t mapq_tab_n;         // coeff_fac / log(l) for l < mapq_tab_n (libm on the host), mapq.hpp:146-184
    int32_t min_len, smatch, smismatch, pad;
    uint64_t* txt_pool; uint64_t txt_cap;                // in 8-byte words; cursor = cursors[15]
};

struct ak_args_t {
    ac_params_t P;
    dp_launch_t D;                           // scoring + reads/text pointers (order/tasks/results unused)
    const moni_mem_t* mems;
    const uint64_t* occs;
    const uint64_t* read_mem_off;
    const uint64_t* offs;
    const int32_t* min_score_of_len;         // 20 + 8*log(l), computed on the host (libm) per read length
    uint32_t max_len;
    uint64_t read_lo, n_reads;               // this launch takes reads [read_lo, read_lo + n_reads) of the resident batch,
    const uint32_t* n_reads_dev;             // when set: the number of reads is read from here (a list filled by earlier kernels of the stream)
    const uint32_t* read_list;               // or, when set, the n_reads reads it names (the reads the staged kernels hand over); records go to recs[read - read_lo]
    ak_slot_t* slots;                        // gridDim.x * AK_NL
    ak_wave_t* waves;                        // gridDim.x
    moni_aln_rec_t* recs;
    uint32_t* cig_pool; uint64_t cig_cap;
    moni_alt_t* alt_pool; uint64_t alt_cap;
    uint64_t* md_pool; uint64_t md_cap;          // in 8-byte words
    ak_fmt_t fmt;                                // SAM text in the kernel (txt_pool == nullptr: records only, the host formats)
    uint64_t* dev_len; uint64_t* dev_off;    // per record slot, in device memory: length of the read's SAM line in bytes and where it is in the text pool
    unsigned long long* dev_sum;             // (8-byte words); dev_sum[0] += records that need the host (no kernel text), dev_sum[1] += aligned reads
    unsigned long long* cursors;             // [0] cigar pool, [1] alt pool, [2] DP problems run, [3] their cells, [4] next read, [5..7] wave cycles: serial
                                             // phases (init+first drive, later drives), dp, [8] DP problems answered from the per-read memo, [9] their cells,
                                             // [14] MD pool (words)
};

// What the record / text writers read of a finished read: a view, so that every align kernel (the persistent per-lane
// state machines below, the staged kernels of align_fast.hip) ends in the same code.
struct ak_final_t {
    uint64_t off; uint32_t m, strand;
    uint64_t ref_pos; int32_t score, score2;
    const uint32_t* cigar; uint32_t n_cigar;
    const uint64_t* alt_pos; const int32_t* alt_score; uint32_t n_alt;
    uint32_t aligned, overflow;
};
__device__ __forceinline__ ak_final_t ak_view(const ac_ws_t& W) {
    ak_final_t V;
    V.off = W.off; V.m = W.m; V.strand = W.fill.strand; V.ref_pos = W.fill.ref_pos; V.score = W.fill.score; V.score2 = W.score2;
    V.cigar = W.cigar; V.n_cigar = W.n_cigar; V.alt_pos = W.alt_pos; V.alt_score = W.alt_score; V.n_alt = W.n_alt;
    V.aligned = W.aligned; V.overflow = W.overflow;
    return V;
}

// MD:Z text and NM of the final alignment (write_MD_core, sam.hpp:249-287), lane-private: read and text bytes through one-word
// register caches, text staged in the slot.  Returns the length, or -1 when it does not fit AK_MD_CAP.
// cig / n_cig / t0: the CIGAR and the text position of its first column; md == nullptr: NM only.
__device__ __attribute__((noinline)) int ak_md(const ak_args_t& A, const ak_final_t& W, const uint32_t* __restrict__ cig, uint32_t n_cig, uint64_t t0,
                                               uint8_t* __restrict__ md, int32_t& nm_out) {
    const uint8_t* __restrict__ text = A.D.text;
    const uint8_t* __restrict__ reads = A.D.reads;
    const uint64_t n_text = A.D.n_text, off = W.off;
    const uint32_t m = W.m, strand = W.strand;
    // byte streams through two-word register caches: the word after the current one is requested as soon as the current one is
    // entered, so its latency overlaps the comparison of eight bases (dir = +1 text and forward reads, -1 reverse-strand reads)
    struct stream_t { const uint8_t* base; uint64_t w, word, nw, nword; };
    auto get = [](stream_t& c, uint64_t a, long dir) -> uint32_t {
        const uint64_t w = a >> 3;
        if (w != c.w) {
            if (w == c.nw) c.word = c.nword; else c.word = *reinterpret_cast<const uint64_t*>(c.base + (w << 3));
            c.w = w;
            if (dir < 0 && w == 0) c.nw = ~0ull;                  // nothing before the first word (buffers are padded at the end only)
            else { c.nw = (uint64_t)((long long)w + dir); c.nword = *reinterpret_cast<const uint64_t*>(c.base + (c.nw << 3)); }
        }
        return (uint32_t)(c.word >> (8 * (a & 7))) & 0xFFu;
    };
    stream_t ts; ts.base = text; ts.w = ts.nw = ~0ull; ts.word = ts.nword = 0;
    stream_t rs; rs.base = reads; rs.w = rs.nw = ~0ull; rs.word = rs.nword = 0;
    auto tb = [&](uint64_t a) -> uint32_t { return dp_nt4(a < n_text ? get(ts, a, 1) : 0u); };
    auto qb = [&](uint32_t k) -> uint32_t {
        if (!strand) return dp_nt4(get(rs, off + k, 1));
        uint32_t b = get(rs, off + m - 1 - k, -1);                                // compl_of (kpbseq.h:120-137), then nt4
        const uint32_t u = b & 0xDFu;
        b = u == 'A' ? 'T' : u == 'C' ? 'G' : u == 'G' ? 'C' : u == 'T' ? 'A' : b;
        return dp_nt4(b);
    };
    int n = 0, NM = 0, l_MD = 0;
    bool ovf = false;
    auto putc = [&](uint8_t ch) { if (!md) return; if (n < AK_MD_CAP) md[n++] = ch; else ovf = true; };
    auto puti = [&](int v) { char b[12]; int k = 0; unsigned u = (unsigned)v; do { b[k++] = (char)('0' + u % 10); u /= 10; } while (u); while (k) putc((uint8_t)b[--k]); };
    uint64_t t = t0; uint32_t q = 0;
    for (uint32_t i = 0; i < n_cig; ++i) {
        const int op = cig[i] & 0xf, len = (int)(cig[i] >> 4);
        if (op == 0) {
            for (int j = 0; j < len; ++j) {
                const uint32_t tc = tb(t + j);
                if (qb(q + j) != tc) { puti(l_MD); putc((uint8_t)"ACGTN"[tc]); l_MD = 0; ++NM; }
                else ++l_MD;
            }
            q += len; t += len;
        } else if (op == 1) { q += len; NM += len; }
        else if (op == 2) {
            puti(l_MD); putc('^');
            for (int j = 0; j < len; ++j) putc((uint8_t)"ACGTN"[tb(t + j)]);
            l_MD = 0; t += len; NM += len;
        }
    }
    if (l_MD > 0) puti(l_MD);
    nm_out = NM;
    return ovf ? -1 : n;
}

// One SAM line (sam.hpp:144-188 as Aligner::emit_record spells it), lane-private, into the slot's text staging; MD text and NM
// come from ak_md.  Returns the length or -1 when it does not fit AK_TXT_CAP.
struct ak_writer_t {                          // eight characters per store
    uint64_t* p; uint64_t acc; int n; bool ovf;
    __device__ __forceinline__ void c(uint8_t ch) {
        acc |= (uint64_t)ch << ((n & 7) * 8);
        if (((++n) & 7) == 0) { if (n <= AK_TXT_CAP) p[(n >> 3) - 1] = acc; else ovf = true; acc = 0; }
    }
    __device__ __forceinline__ void flush() { if (n & 7) { if (n < AK_TXT_CAP) p[n >> 3] = acc; else ovf = true; } }
    __device__ __forceinline__ void lit(const char* q) { while (*q) c((uint8_t)*q++); }
    __device__ __forceinline__ void i(int v) {
        char b[12]; int k = 0; unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
        do { b[k++] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) c('-');
        while (k) c((uint8_t)b[--k]);
    }
};
// bytes [a, a + len) of a padded device buffer (or the same backwards), through one-word register caches
struct ak_reader_t {
    const uint8_t* base; uint64_t w, word;
    __device__ __forceinline__ uint8_t at(uint64_t a) {
        const uint64_t x = a >> 3;
        if (x != w) { word = *reinterpret_cast<const uint64_t*>(base + (x << 3)); w = x; }
        return (uint8_t)(word >> (8 * (a & 7)));
    }
};

__device__ __forceinline__ uint8_t ak_compl(uint8_t b) {        // kpbseq.h:120-137
    const uint8_t u = b & 0xDFu;
    return u == 'A' ? 'T' : u == 'C' ? 'G' : u == 'G' ? 'C' : u == 'T' ? 'A' : b;
}

// lcig / n_lcig / lifted: the alignment lifted to the reference contig (columns 3, 4, 6, MD, NM); W.cigar / W.ref_pos: the
// alignment on the pangenome text (OA tag, with lift_nm); aligner_ksw2.hpp:3116-3175
__device__ __attribute__((noinline)) int ak_emit(const ak_args_t& A, const ak_final_t& W, uint64_t r, bool aligned, const uint8_t* md, int md_len, int32_t nm,
                                                 int32_t lift_nm, const uint32_t* __restrict__ lcig, uint32_t n_lcig, uint64_t lifted, uint8_t* __restrict__ out) {
    const ak_fmt_t& F = A.fmt;
    ak_writer_t w; w.p = reinterpret_cast<uint64_t*>(out); w.acc = 0; w.n = 0; w.ovf = false;
    const uint64_t rd = W.off;                                   // offsets into the (padded) device buffers
    const bool ql = F.quals != nullptr;
    const uint32_t m = W.m;
    ak_reader_t R_reads{A.D.reads, ~0ull, 0}, R_quals{F.quals, ~0ull, 0}, R_rn{F.rnames, ~0ull, 0}, R_sn{F.snames, ~0ull, 0};
    auto put = [&](ak_reader_t& R, uint64_t a0, size_t len) { for (size_t k = 0; k < len; ++k) w.c(R.at(a0 + k)); };
    put(R_rn, F.rname_off[r], (size_t)(F.rname_off[r + 1] - F.rname_off[r]));
    if (!aligned) {
        w.lit("\t4\t*\t0\t255\t*\t*\t0\t0\t");
        put(R_reads, rd, m); w.c('\t');
        if (ql) put(R_quals, rd, m); else w.c('*');
        w.c('\n');
        w.flush();
        return w.ovf ? -1 : w.n;
    }
    const uint32_t strand = W.strand;
    const int32_t score = W.score, score2 = W.score2;
    uint64_t ref_len = 0;
    for (uint32_t k = 0; k < n_lcig; ++k) { const int op = lcig[k] & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += lcig[k] >> 4; }
    const uint64_t rk = ac_rank1(A.P, W.ref_pos + 1);
    const uint32_t sid = (uint32_t)(rk - 1);
    const int oa_pos = (int)(W.ref_pos - A.P.seq_starts[rk - 1] + 1);         // sam->lift_pos: the position on the pangenome sequence
    const uint64_t lrk = ac_rank1(A.P, lifted + 1);
    const uint32_t lsid = (uint32_t)(lrk - 1);
    const int pos1 = (int)(lifted - A.P.seq_starts[lrk - 1] + 1);                   // sam->pos
    const bool mapped = ref_len > 0;
    // compute_mapq_se_bwa (mapq.hpp:146-184), the operations in the host's order, none contracted
    int mapq = 0;
    {
        const int32_t rl = mapped ? (int32_t)ref_len : 0;
        const int32_t l = rl > (int32_t)m ? rl : (int32_t)m;
        const int32_t sub = score2 ? score2 : F.min_len * F.smatch;
        if (sub < score) {
            const double identity = __dsub_rn(1., __ddiv_rn(__ddiv_rn((double)(l * F.smatch - score), (double)(F.smatch + F.smismatch)), (double)l));
            if (score != 0) {
                double tmp = (double)l < 50.0 ? 1. : ((uint32_t)l < F.mapq_tab_n ? F.mapq_tab[l] : F.mapq_tab[F.mapq_tab_n - 1]);
                tmp = __dmul_rn(tmp, __dmul_rn(identity, identity));
                const double v = __dadd_rn(__dmul_rn(__dmul_rn(__ddiv_rn(__dmul_rn(6.02, (double)(score - sub)), (double)F.smatch), tmp), tmp), .499);
                mapq = (int)v;
            }
            if (mapq > 60) mapq = 60;
            if (mapq < 0) mapq = 0;
            mapq = (int)__dadd_rn(__dmul_rn((double)mapq, 1.), .499);
        }
    }
    const uint64_t sn = F.sname_off[sid];
    const size_t sn_len = F.sname_off[sid + 1] - F.sname_off[sid];
    w.c('\t'); w.i(strand ? 16 : 0); w.c('\t');
    if (mapped) put(R_sn, F.sname_off[lsid], (size_t)(F.sname_off[lsid + 1] - F.sname_off[lsid])); else w.c('*');
    w.c('\t'); w.i(mapped ? pos1 : 0); w.c('\t'); w.i(mapq); w.c('\t');
    if (mapped) { for (uint32_t k = 0; k < n_lcig; ++k) { w.i((int)(lcig[k] >> 4)); w.c((uint8_t)"MID"[lcig[k] & 0xf]); } } else w.c('*');
    w.lit("\t*\t0\t0\t");
    if (strand) for (uint32_t k = 0; k < m; ++k) w.c(ak_compl(R_reads.at(rd + m - 1 - k))); else put(R_reads, rd, m);
    w.c('\t');
    if (ql) { if (strand) for (uint32_t k = 0; k < m; ++k) w.c(R_quals.at(rd + m - 1 - k)); else put(R_quals, rd, m); } else w.c('*');
    w.lit("\tAS:i:"); w.i(score); w.lit("\tNM:i:"); w.i(mapped ? nm : 0);
    if (score2 != 0) { w.lit("\tZS:i:"); w.i(score2); }
    w.lit("\tMD:Z:"); if (mapped) for (int k = 0; k < md_len; ++k) w.c(md[k]);
    w.lit("\tOA:Z:"); put(R_sn, sn, sn_len); w.c(','); w.i(oa_pos); w.lit(strand ? ",-," : ",+,");
    for (uint32_t k = 0; k < W.n_cigar; ++k) { w.i((int)(W.cigar[k] >> 4)); w.c((uint8_t)"MID"[W.cigar[k] & 0xf]); }
    w.c(','); w.i(mapq); w.c(','); w.i(lift_nm); w.c(';');
    w.lit("\tAA:Z:");
    for (uint32_t k = 0; k < W.n_alt; ++k) {
        const uint64_t rk2 = ac_rank1(A.P, W.alt_pos[k] + 1);
        const uint32_t s2 = (uint32_t)(rk2 - 1);
        put(R_sn, F.sname_off[s2], (size_t)(F.sname_off[s2 + 1] - F.sname_off[s2]));
        w.c(','); w.i((int)(W.alt_pos[k] - A.P.seq_starts[rk2 - 1] + 1)); w.c(','); w.i(W.alt_score[k]); w.c(';');
    }
    w.c('\n');
    w.flush();
    return w.ovf ? -1 : w.n;
}

// the record of a finished read (lane-private)
__device__ __attribute__((noinline)) void ak_write_record(const ak_args_t& A, const ak_final_t& W, uint64_t* __restrict__ md_tmp, uint64_t* __restrict__ txt_tmp,
                                                          uint32_t* __restrict__ lcig, uint32_t lcig_cap, uint64_t slot_in_launch, uint64_t read_index) {
    moni_aln_rec_t rec;
    rec.status = W.overflow ? 2u : (W.aligned ? 1u : 0u);
    rec.strand = W.strand; rec.ref_pos = W.ref_pos; rec.score = W.score; rec.score2 = W.score2;
    rec.n_cigar = 0; rec.n_alt = 0; rec.cigar_off = 0; rec.alt_off = 0; rec.nm = 0; rec.md_len = 0; rec.md_off = 0;
    rec.txt_len = 0; rec.lift_nm = 0; rec.txt_off = 0;
    int32_t nm = 0, lift_nm = 0;
    int md_len = 0, n_lcig = 0;
    uint64_t lifted = 0;
    if (rec.status == 1) {
        // the alignment lifted to the reference contig (aligner_ksw2.hpp:3133-3160): CIGAR, position, MD / NM over the lifted window;
        // NM of the alignment on the pangenome text stays for the OA tag
        const uint64_t rk = ac_rank1(A.P, W.ref_pos + 1);
        const moni_lift_seq_t L = A.P.lift_seqs[rk - 1];
        const moni_lift_run_t* __restrict__ runs = A.P.lift_runs + L.run_off;
        const uint64_t start = W.ref_pos - A.P.seq_starts[rk - 1];
        n_lcig = lift_cigar(runs, L.n_runs, start, W.cigar, W.n_cigar, lcig, lcig_cap);
        lifted = L.second + lift_pos(runs, L.n_runs, start);
        bool same = n_lcig == (int)W.n_cigar && lifted == W.ref_pos;
        for (uint32_t k = 0; same && k < W.n_cigar; ++k) same = lcig[k] == W.cigar[k];
        uint64_t lref = 0;
        for (int k = 0; k < n_lcig; ++k) { const int op = lcig[k] & 0xf; if (op == 0 || op == 2 || op == 3) lref += lcig[k] >> 4; }
        if (n_lcig >= 0 && lref > 0) md_len = ak_md(A, W, lcig, (uint32_t)n_lcig, lifted, reinterpret_cast<uint8_t*>(md_tmp), nm);
        if (same) lift_nm = nm; else (void)ak_md(A, W, W.cigar, W.n_cigar, W.ref_pos, nullptr, lift_nm);
        const unsigned long long md_words = md_len > 0 ? (unsigned long long)((md_len + 7) >> 3) : 0ull;
        const unsigned long long co = atomicAdd(&A.cursors[0], (unsigned long long)W.n_cigar);
        const unsigned long long ao = atomicAdd(&A.cursors[1], (unsigned long long)W.n_alt);
        const unsigned long long mo = atomicAdd(&A.cursors[14], md_words);
        if (md_len < 0 || n_lcig < 0 || co + W.n_cigar > A.cig_cap || ao + W.n_alt > A.alt_cap || mo + md_words > A.md_cap) rec.status = 2;      // pool too small: let the host pipeline redo the read
        else {
            rec.n_cigar = W.n_cigar; rec.cigar_off = co; rec.n_alt = W.n_alt; rec.alt_off = ao;
            rec.nm = nm; rec.lift_nm = lift_nm; rec.md_len = (uint32_t)md_len; rec.md_off = mo;
            for (uint32_t k = 0; k < W.n_cigar; ++k) A.cig_pool[co + k] = W.cigar[k];
            for (uint32_t k = 0; k < W.n_alt; ++k) { moni_alt_t x; x.pos = W.alt_pos[k]; x.score = W.alt_score[k]; x.pad = 0; A.alt_pool[ao + k] = x; }
            for (unsigned long long k = 0; k < md_words; ++k) A.md_pool[mo + k] = md_tmp[k];
        }
    }
    if (A.fmt.txt_pool && rec.status != 2) {
        // the finished SAM line; when it does not fit (staging or pool) the host formats this record from the fields above
        const int n = ak_emit(A, W, read_index, rec.status == 1, reinterpret_cast<const uint8_t*>(md_tmp), md_len, nm, lift_nm, lcig,
                              (uint32_t)(n_lcig > 0 ? n_lcig : 0), lifted, reinterpret_cast<uint8_t*>(txt_tmp));
        if (n > 0) {
            const unsigned long long words = (unsigned long long)((n + 7) >> 3);
            const unsigned long long to = atomicAdd(&A.cursors[15], words);
            if (to + words <= A.fmt.txt_cap) {
                for (unsigned long long k = 0; k < words; ++k) A.fmt.txt_pool[to + k] = txt_tmp[k];
                rec.txt_len = (uint32_t)n; rec.txt_off = to;
            }
        }
    }
    A.recs[slot_in_launch] = rec;
    if (A.dev_len) {
        A.dev_len[slot_in_launch] = rec.txt_len; A.dev_off[slot_in_launch] = rec.txt_off;
        if (rec.status == 2 || rec.txt_len == 0) atomicAdd(&A.dev_sum[0], 1ull);
        else if (rec.status == 1) atomicAdd(&A.dev_sum[1], 1ull);
    }
}

extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
align_kernel(const ak_args_t A) {
    __shared__ dp_lds_t L;
    __shared__ moni_dp_task_t s_tasks[AC_MAX_TASKS];
    __shared__ unsigned long long s_cnt[8];
    __shared__ unsigned long long s_hist[8];
    __shared__ unsigned long long s_cy[8];        // wave cycles inside phase 2: per-read setup, memo lookups, DP by live-row class (<=16, <=32, <=64, >64)       // statistics: DP problems run, cells, memo hits, their cells, wave cycles in the phases
    enum { C_DP = 0, C_CELLS, C_MEMO, C_MEMO_CELLS, C_INIT, C_DRIVE, C_CYDP };
    const int lane = threadIdx.x;
    if (A.n_reads_dev && *A.n_reads_dev == 0) return;        // nothing was handed over by the staged kernels
    if (lane < 8) { s_cnt[lane] = 0; s_hist[lane] = 0; s_cy[lane] = 0; }
    __syncthreads();
    ak_slot_t* __restrict__ S = A.slots + (size_t)blockIdx.x * AK_NL + (lane < AK_NL ? lane : 0);
    ac_ws_t& W = S->ws;
    uint8_t* __restrict__ dirs = A.waves[blockIdx.x].dirs;
    if (lane < AK_NL) for (int k = 0; k < 6; ++k) W.prof[k] = 0;
    int state = lane < AK_NL ? 0 : 2;             // 0: wants a read, 1: waits for DP results, 2: no more reads, 3: finished, record not yet written
    uint64_t r_in = 0;                            // read index inside the launch
    uint64_t r_slot = 0, r_read = 0;              // its record slot and its index in the resident batch
    while (true) {
        // ---- phase 1 (lane-private): take a read; seeds -> chains -> first DP request ----
        const long long c0 = clock64();
        const bool start = __popcll(__ballot(state == 0 || state == 3)) >= AK_START_MIN || __ballot(state == 1) == 0ull;
        // finished reads write their records (MD/NM, pool entries) together, like the starts: it is lane-private work too
        if (state == 3 && start) { ak_write_record(A, ak_view(W), S->md_tmp, S->txt_tmp, S->lcig, AC_MAX_CIGAR, r_slot, r_read); state = 0; }
        if (state == 0 && start) {
            r_in = atomicAdd(&A.cursors[4], 1ull);
            if (r_in >= (A.n_reads_dev ? (uint64_t)*A.n_reads_dev : A.n_reads)) state = 2;
            else {
                const uint64_t r = A.read_list ? (uint64_t)A.read_list[r_in] : A.read_lo + r_in;
                r_read = r; r_slot = r - A.read_lo;
                W.off = A.offs[r]; W.m = (uint32_t)(A.offs[r + 1] - A.offs[r]);
                W.min_score = A.min_score_of_len[W.m <= A.max_len ? W.m : A.max_len];
                S->memo_n = 0;
                bool chained = false;
                if (W.m >= 4096) {                      // far beyond what the kernel's DP takes, and the chaining nodes keep read coordinates in 16 bits
                    W.stage = AC_DONE; W.aligned = 0; W.overflow = 1; W.n_cigar = 0; W.n_tasks = 0; W.n_alt = 0; W.score2 = 0;
                } else chained = ac_init(W, A.P, A.mems, A.read_mem_off[r], A.read_mem_off[r + 1], A.occs);
                if (chained) ac_drive(W, A.P, nullptr, nullptr);
                if (chained && !W.overflow && W.stage != AC_DONE) state = 1;
                else ak_write_record(A, ak_view(W), S->md_tmp, S->txt_tmp, S->lcig, AC_MAX_CIGAR, r_slot, r_read);
            }
        }
        const unsigned long long waiting = __ballot(state == 1);
        const long long c1 = clock64();
        if (lane == 0) s_cnt[C_INIT] += (unsigned long long)(c1 - c0);
        if (waiting == 0ull) { if (__ballot(state == 0 || state == 3) == 0ull) break; continue; }
        __threadfence();
        // ---- phase 2 (whole wave): the DP problems of every waiting read, one read after the other ----
        for (unsigned long long todo = waiting; todo; todo &= todo - 1) {
            const int src = __ffsll((long long)todo) - 1;
            const long long y0 = clock64();
            ak_slot_t* __restrict__ Q = A.slots + (size_t)blockIdx.x * AK_NL + src;
            // everything phase 2 needs of this read in one round trip: its DP requests (all AC_MAX_TASKS slots, the count decides
            // later) and its memo (lane e holds entry e: key, target offset, the three result words)
            const uint32_t* __restrict__ tsrc = reinterpret_cast<const uint32_t*>(Q->ws.tasks);
            constexpr uint32_t TW = AC_MAX_TASKS * (uint32_t)(sizeof(moni_dp_task_t) / 4);
            constexpr uint32_t TWL = (TW + 63) / 64;                 // words per lane
            uint32_t tw[TWL];
#pragma unroll
            for (uint32_t x = 0; x < TWL; ++x) tw[x] = (uint32_t)lane + 64 * x < TW ? tsrc[lane + 64 * x] : 0u;
            const uint32_t nt = Q->ws.n_tasks;
            const uint64_t read_off = Q->ws.off;
            uint32_t memo_n = Q->memo_n;
            uint64_t mk = lane < AK_MEMO ? Q->memo_key[lane] : ~0ull;
            uint64_t mt = lane < AK_MEMO ? Q->memo_toff[lane] : 0ull;
            dp_brief_t mv = Q->memo_val[lane < AK_MEMO ? lane : 0];
            __syncthreads();
#pragma unroll
            for (uint32_t x = 0; x < TWL; ++x) if ((uint32_t)lane + 64 * x < TW) ((uint32_t*)s_tasks)[lane + 64 * x] = tw[x];
            if ((uint32_t)lane >= memo_n) mk = ~0ull;
            __syncthreads();
            bool too_big = false;
            uint32_t cig_used = 0;
            if (lane == 0) s_cy[0] += (unsigned long long)(clock64() - y0);
            for (uint32_t t = 0; t < nt; ++t) {
                const moni_dp_task_t task = s_tasks[t];
                const bool with_cigar = !(task.flag & DP_EZ_SCORE_ONLY);
                const uint32_t cig_need = with_cigar && task.qlen > 0 && task.tlen > 0 ? (uint32_t)(task.qlen + task.tlen + 2) : 0u;
                if (task.qlen > DP_LDS_Q || task.tlen > DP_LDS_T || cig_used + cig_need > AK_CIG_CAP ||
                    (with_cigar && (uint64_t)(task.qlen + task.tlen - 1) * (uint64_t)task.tlen > AK_DIRS_CAP)) { too_big = true; break; }
                const uint32_t cig_at = cig_used;
                cig_used += cig_need;
                uint32_t* cg = Q->cig + cig_at;
                const unsigned long long cells = (unsigned long long)(task.qlen > 0 ? task.qlen : 0) * (unsigned long long)(task.tlen > 0 ? task.tlen : 0);
                // memo: score-only problems on the index text (every problem of the chain-selection loop)
                const bool memoable = !with_cigar && (task.reserved & DP_T_TEXT) && (task.reserved & DP_Q_READS) && cells > 0;
                const uint64_t key = ((task.q_off - read_off) & 0xFFFFull) | ((uint64_t)(uint32_t)task.qlen & 0xFFFFull) << 16 |
                                     ((uint64_t)(uint32_t)task.tlen & 0xFFFFull) << 32 | ((uint64_t)task.flag & 0xFFull) << 48 | ((uint64_t)task.reserved & 0xFFull) << 56;
                // one query base against one target base, globally: when the diagonal move beats both gap moves the result is
                // that substitution score and "1M" (the gap between two MEMs that a single mismatch separates: the commonest problem)
                if (task.qlen == 1 && task.tlen == 1 && !(task.flag & DP_EZ_EXTZ_ONLY) && (task.reserved & DP_Q_READS) && (task.reserved & DP_T_TEXT)) {
                    uint32_t qc = dp_nt4(A.D.reads[task.q_off]);
                    if ((task.reserved & DP_Q_COMP) && qc < 4) qc = 3 - qc;
                    const uint32_t tc = dp_nt4(task.t_off < A.D.n_text ? A.D.text[task.t_off] : 0u);
                    const int32_t z = (tc == (uint32_t)A.D.wild || qc == (uint32_t)A.D.wild) ? A.D.sc_N : (tc == qc ? A.D.sc_mch : A.D.sc_mis);
                    const int32_t gap = dp_bound(0, A.D.qo, A.D.e) - A.D.qo - A.D.e;        // E(0,0) = F(0,0)
                    if (z > gap) {
                        if (lane == 0) {
                            moni_dp_result_t x;
                            x.max = z > 0 ? z : 0; x.max_q = x.max_t = z > 0 ? 0 : -1;
                            x.mqe = z; x.mqe_t = 0; x.mte = z; x.mte_q = -15; x.score = z;      // mte_q = r - ((tlen-1+16)/16*16 - 1) at r = 0
                            x.reach_end = 0; x.zdropped = 0; x.n_cigar = with_cigar ? 1u : 0u; x.cigar_off = cig_at;
                            if (with_cigar) cg[0] = 1u << 4;
                            Q->res[t] = x;
                            s_cy[6]++;
                        }
                        continue;
                    }
                }
                int hit = -1;
                const long long y1 = clock64();
                if (memoable) {
                    for (unsigned long long cand = __ballot(mk == key); cand && hit < 0; cand &= cand - 1) {
                        const int e = __ffsll((long long)cand) - 1;
                        const uint64_t toff_e = ((uint64_t)(uint32_t)__shfl((int)(mt >> 32), e) << 32) | (uint64_t)(uint32_t)__shfl((int)(mt & 0xFFFFFFFFull), e);
                        if (toff_e == task.t_off || ak_same_target(A.D, (int)task.reserved, toff_e, task.t_off, task.tlen)) hit = e;
                    }
                }
                const long long y2 = clock64();
                if (lane == 0) s_cy[1] += (unsigned long long)(y2 - y1);
                if (hit >= 0) {
                    if (lane == hit) {          // the lane that holds the entry writes the result: no load on this path
                        moni_dp_result_t x;
                        x.max = 0; x.max_q = x.max_t = -1; x.mqe = mv.mqe; x.mqe_t = mv.mqe_t; x.mte = DP_NEG_INF; x.mte_q = -1; x.score = mv.score;
                        x.reach_end = 0; x.zdropped = 0; x.n_cigar = 0; x.cigar_off = cig_at;
                        Q->res[t] = x;
                    }
                    if (lane == 0) { s_cnt[C_MEMO]++; s_cnt[C_MEMO_CELLS] += cells; }
                } else {
                    const dp_brief_t br = extz_wave_lds_lite(A.D, task, L, dirs, cg, &Q->res[t], cig_at);
                    if (lane == 0) {
                        s_cnt[C_DP]++; s_cnt[C_CELLS] += cells;
                        { const int lr = task.qlen < task.tlen ? task.qlen : task.tlen; const int b = lr <= 16 ? 0 : lr <= 32 ? 1 : lr <= 64 ? 2 : 3; s_hist[b]++; s_hist[4 + b] += cells; }
                        if (memoable && memo_n < AK_MEMO) { Q->memo_key[memo_n] = key; Q->memo_toff[memo_n] = task.t_off; Q->memo_val[memo_n] = br; }
                    }
                    if (memoable && memo_n < AK_MEMO) { if ((uint32_t)lane == memo_n) { mk = key; mt = task.t_off; mv = br; } ++memo_n; }
                    __syncthreads();
                    if (lane == 0) { const int lr = task.qlen < task.tlen ? task.qlen : task.tlen; s_cy[2 + (lr <= 16 ? 0 : lr <= 32 ? 1 : lr <= 64 ? 2 : 3)] += (unsigned long long)(clock64() - y2); }
                }
            }
            if (lane == 0) { Q->memo_n = memo_n; if (too_big) Q->ws.overflow = 1; }
        }
        __threadfence();
        __syncthreads();
        const long long c2 = clock64();
        if (lane == 0) s_cnt[C_CYDP] += (unsigned long long)(c2 - c1);
        // ---- phase 3 (lane-private): results -> next DP request, or the finished record ----
        if (state == 1) {
            if (!W.overflow) ac_drive(W, A.P, S->res, S->cig);
            if (W.overflow || W.stage == AC_DONE) state = 3;
        }
        if (lane == 0) s_cnt[C_DRIVE] += (unsigned long long)(clock64() - c2);
    }
    __syncthreads();
    if (lane < AK_NL) for (int k = 0; k < 4; ++k) atomicAdd(&A.cursors[10 + k], W.prof[k]);
    if (lane == 0) {
        atomicAdd(&A.cursors[2], s_cnt[C_DP]); atomicAdd(&A.cursors[3], s_cnt[C_CELLS]); atomicAdd(&A.cursors[5], s_cnt[C_INIT]); atomicAdd(&A.cursors[6], s_cnt[C_DRIVE]);
        atomicAdd(&A.cursors[7], s_cnt[C_CYDP]);
        for (int k = 0; k < 8; ++k) atomicAdd(&A.cursors[16 + k], s_hist[k]);
        for (int k = 0; k < 7; ++k) atomicAdd(&A.cursors[24 + k], s_cy[k]); atomicAdd(&A.cursors[8], s_cnt[C_MEMO]); atomicAdd(&A.cursors[9], s_cnt[C_MEMO_CELLS]);
    }
}
