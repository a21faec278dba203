// pe_lines.hip - the two SAM lines of a pair written on the GPU: what the host finishing (pe_host.hpp: pe_emit_fast) does per pair -
// per mate the lift-over of position and CIGAR, MD / NM of the lifted and of the unlifted alignment, the single-end MAPQ
// (aligner_ksw2.hpp:3133-3175, sam.hpp:249-287, mapq.hpp:146-184), then the tail of paired_chain_score (aligner_ksw2.hpp:2200-2288: PNEXT / TLEN /
// flags / compute_mapq_pe_bwa, or one mate placed by the other), remove_slash_mate, the two lines (sam.hpp:144-188) - in the form of
// finish_wave_kernel: ONE WAVE PER PAIR, the line listed as segments by lane 0 and rendered by all lanes.  Input: the pair's pe_rec_t with its
// CIGARs and alternatives in the pools, whoever wrote them (pe_finish_kernel, or pe_align_kernel for the pairs handed over).  Lines that do not
// fit the staging (CIGAR, MD, segments, line length, more than PEL_MAX_ALT alternatives, a pair the kernels gave up on) are counted in dev_sum[0]:
// the host then finishes that chunk (pe_host.hpp) - same bytes either way.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// (included by moni_hip.hip after align_fast.hip, pe_kernel.hip and pe_fast.hip)

#define PEL_MAX_ALT 16
#define PEL_PAIR_WORDS (2 * (AFS_LINE / 8))

struct pel_args_t {
    ac_params_t P;                    // lift tables, sequence starts
    dp_launch_t D;                    // reads, text
    ak_fmt_t F;                       // names of the 2 N reads, qualities, sequence names, the coeff_fac / log(l) table, min_len / smatch / smismatch
    const uint64_t* offs;             // of the 2 N reads
    const pe_rec_t* recs; const uint32_t* cig_pool; const moni_alt_t* alt_pool;      // of the chunk
    uint64_t pair_lo, n_pairs;        // the chunk
    const int32_t* subn_tab; uint32_t subn_tab_n;      // (int)(4.343 * log(sub_n + 1) + .499), libm on the host (mapq.hpp:176, 208)
    const int32_t* min_score_of_len; uint32_t max_len; // 20 + 8 log(l)
    uint64_t* txt_pool;               // PEL_PAIR_WORDS 8-byte words per pair of the chunk: no cursor, a pair's lines lie at its own place
    uint64_t* dev_len; uint64_t* dev_off;              // per line (2 i + k): bytes, word offset in the pool
    unsigned long long* dev_sum;      // [0] pairs the host has to finish, [8 + 8 s] aligned pairs (16 shards)
};

struct pel_mate_t {
    uint8_t seq[AF_MAX_READ];         // the mate as SEQ prints it (reverse-complemented when the line says so): also the MD walk's query
    uint32_t cig[AFS_CIG], lcig[AFS_LCIG];
    uint32_t md_item[AFS_MAXMD]; uint16_t md_off[AFS_MAXMD + 1];
    uint16_t cig_off[AFS_CIG + 1], lcig_off[AFS_LCIG + 1];
    uint32_t alt_sid[PEL_MAX_ALT], alt_p1[PEL_MAX_ALT]; int32_t alt_score[PEL_MAX_ALT];
    uint32_t n_lcig, ovf;
    uint64_t lifted;
};
struct pel_wave_t {
    uint8_t line[AFS_LINE];
    pel_mate_t M[2];
    pe_rec_t rec;
    uint16_t seg_off[AFS_MAXSEG + 1]; uint8_t seg_kind[AFS_MAXSEG]; uint32_t seg_val[AFS_MAXSEG];
    uint32_t n_seg, total, seq_at, qual_at;
    uint8_t names[AFW_NAMES]; uint16_t name_off[AFW_NSEQ + 2];
};

// what pe_emit_fast keeps per mate (PeMateFin), in registers (uniform over the wave)
struct pel_fin_t {
    bool filled, unmapped_lft, cigar_star;
    uint32_t flag; uint64_t pos, mapq, pnext; int32_t as, nm, zs; long long tlen;
    int rname, lift_rname; uint64_t lift_pos; int32_t lift_nm; uint32_t rlen, n_md, n_cig, n_lcig;
};

// MH_RAW_MAPQ (mapq.hpp:144): (int)(6.02 * diff / a + .499)
__device__ __forceinline__ int pel_raw_mapq(int32_t diff, int32_t a) { return (int)__dadd_rn(__ddiv_rn(__dmul_rn(6.02, (double)diff), (double)a), .499); }

// compute_mapq_se_bwa with sub_n (mapq.hpp:146-184), the host's operation order, nothing contracted
__device__ __forceinline__ uint64_t pel_mapq_se(const pel_args_t& X, int32_t score, int32_t score2, int32_t rlen, int32_t qlen, int32_t sub_n, bool& unknown) {
    const ak_fmt_t& F = X.F;
    int32_t mapq = 0;
    const int32_t l = rlen > qlen ? rlen : qlen;
    const int32_t sub = score2 ? score2 : F.min_len * F.smatch;
    if (sub >= score) return 0;
    const double identity = __dsub_rn(1., __ddiv_rn(__ddiv_rn((double)(l * F.smatch - score), (double)(F.smatch + F.smismatch)), (double)l));
    if (score != 0) {
        double tmp = (double)l < 50.0 ? 1. : ((uint32_t)l < F.mapq_tab_n ? F.mapq_tab[l] : F.mapq_tab[F.mapq_tab_n - 1]);
        if ((uint32_t)l >= F.mapq_tab_n) unknown = true;
        tmp = __dmul_rn(tmp, __dmul_rn(identity, identity));
        mapq = (int)__dadd_rn(__dmul_rn(__dmul_rn(__ddiv_rn(__dmul_rn(6.02, (double)(score - sub)), (double)F.smatch), tmp), tmp), .499);
    }
    if (sub_n > 0) { if ((uint32_t)sub_n < X.subn_tab_n) mapq -= X.subn_tab[sub_n]; else unknown = true; }
    if (mapq > 60) mapq = 60;
    if (mapq < 0) mapq = 0;
    mapq = (int)__dadd_rn(__dmul_rn((double)mapq, 1.), .499);
    return (uint64_t)(int64_t)mapq;
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) pe_lines_kernel(const pel_args_t X) {
    __shared__ pel_wave_t L;
    const int lane = threadIdx.x;
    const ak_fmt_t& F = X.F;
    const ac_params_t& P = X.P;
    const dp_launch_t& D = X.D;
    const uint32_t n_seq = (uint32_t)P.n_seq;
    const bool names_lds = n_seq <= AFW_NSEQ && F.sname_off[n_seq] <= AFW_NAMES;
    if (names_lds) {
        for (uint32_t k = lane; k <= n_seq; k += 64) L.name_off[k] = (uint16_t)F.sname_off[k];
        for (uint32_t k = lane; k < F.sname_off[n_seq]; k += 64) L.names[k] = F.snames[k];
    }
#define PEL_NAME_LEN(sid) (names_lds ? (uint32_t)(L.name_off[(sid) + 1] - L.name_off[sid]) : F.sname_off[(sid) + 1] - F.sname_off[sid])
    for (uint64_t r_in = blockIdx.x; r_in < X.n_pairs; r_in += gridDim.x) {
        const uint64_t pair = X.pair_lo + r_in;
        __syncthreads();
        {   // the pair's record with one round trip
            const uint32_t* src = reinterpret_cast<const uint32_t*>(&X.recs[r_in]);
            uint32_t* dst = reinterpret_cast<uint32_t*>(&L.rec);
            for (uint32_t w = lane; w < sizeof(pe_rec_t) / 4; w += 64) dst[w] = src[w];
        }
        const uint64_t off[2] = {X.offs[2 * pair], X.offs[2 * pair + 1]};
        const uint32_t m[2] = {(uint32_t)(off[1] - off[0]), (uint32_t)(X.offs[2 * pair + 2] - off[1])};
        // names: remove_slash_mate (common/sam.hpp:132-141); RNEXT is "=" when the two are the same
        const uint64_t nb[3] = {F.rname_off[2 * pair], F.rname_off[2 * pair + 1], F.rname_off[2 * pair + 2]};
        uint32_t nl[2] = {(uint32_t)(nb[1] - nb[0]), (uint32_t)(nb[2] - nb[1])};
        for (int k = 0; k < 2; ++k)
            if (nl[k] >= 2 && F.rnames[nb[k] + nl[k] - 2] == '/' && (F.rnames[nb[k] + nl[k] - 1] == '1' || F.rnames[nb[k] + nl[k] - 1] == '2')) nl[k] -= 2;
        bool same_name = nl[0] == nl[1];
        if (same_name) {
            bool diff = false;
            for (uint32_t k = lane; k < nl[0]; k += 64) diff = diff || F.rnames[nb[0] + k] != F.rnames[nb[1] + k];
            same_name = __ballot(diff) == 0ull;
        }
        __syncthreads();
        const pe_rec_t& R = L.rec;
        bool to_host = R.status >= 2 || m[0] > AF_MAX_READ || m[1] > AF_MAX_READ;          // status 2: beyond the kernels' capacities (the host pipeline's pair)
        const bool finalized = R.status == 1;
        const uint32_t strand = R.strand;
        const bool rev[2] = {finalized && strand != 0, finalized && strand == 0};
        pel_fin_t s[2];
        for (int k = 0; k < 2; ++k) {
            pel_fin_t& f = s[k];
            f.filled = false; f.unmapped_lft = false; f.cigar_star = true; f.flag = 4; f.pos = 0; f.mapq = 255; f.pnext = 0; f.as = 0; f.nm = 0; f.zs = 0; f.tlen = 0;
            f.rname = -1; f.lift_rname = -1; f.lift_pos = 0; f.lift_nm = 0; f.rlen = 0; f.n_md = 0; f.n_cig = 0; f.n_lcig = 0;
        }
        // the mates as SEQ prints them
        if (!to_host) for (int k = 0; k < 2; ++k)
            for (uint32_t i = lane; i < m[k]; i += 64) L.M[k].seq[i] = rev[k] ? ak_compl(D.reads[off[k] + m[k] - 1 - i]) : D.reads[off[k] + i];
        bool ok[2] = {false, false};
        bool unknown = false;
        if (finalized && !to_host) {
            for (int k = 0; k < 2; ++k) {
                if (!R.filled[k]) continue;
                pel_mate_t& M = L.M[k];
                pel_fin_t& f = s[k];
                f.filled = true;
                const uint64_t ref_pos = R.ref_pos[k];
                const uint32_t n_cig = R.n_cigar[k], n_alt = R.n_alt[k];
                if (n_cig > AFS_CIG || n_alt > PEL_MAX_ALT) { to_host = true; break; }
                uint32_t hint0 = 0xFFFFFFFFu;
                const uint32_t sid0 = ac_seq_of(P, ref_pos, &hint0);
                const moni_lift_seq_t LS0 = P.lift_seqs[sid0];
                for (uint32_t i = lane; i < n_cig; i += 64) M.cig[i] = X.cig_pool[R.cigar_off[k] + i];
                if ((uint32_t)lane < n_alt) {
                    const moni_alt_t a = X.alt_pool[R.alt_off[k] + lane];
                    const uint32_t s2 = ac_seq_of(P, a.pos);
                    M.alt_sid[lane] = s2; M.alt_p1[lane] = (uint32_t)(a.pos - P.lift_seqs[s2].start + 1); M.alt_score[lane] = a.score;
                }
                __syncthreads();
                if (lane == 0) {          // the alignment lifted to the reference contig (aligner_ksw2.hpp:3133-3160)
                    const moni_lift_run_t* __restrict__ runs = P.lift_runs + LS0.run_off;
                    const uint32_t rel = hint0 == 0xFFFFFFFFu ? hint0 : hint0 - LS0.run_off;
                    uint64_t lp = 0;
                    const int nlc = lift_cigar(runs, LS0.n_runs, ref_pos - LS0.start, M.cig, n_cig, M.lcig, AFS_LCIG, rel, &lp);
                    M.ovf = nlc < 0 ? 1u : 0u; M.n_lcig = nlc < 0 ? 0u : (uint32_t)nlc; M.lifted = LS0.second + lp;
                }
                __syncthreads();
                if (M.ovf) { to_host = true; break; }
                const uint32_t n_lcig = M.n_lcig;
                const uint64_t lifted = M.lifted;
                uint64_t ref_len = 0, ref_len_u = 0;
                for (uint32_t i = 0; i < n_lcig; ++i) { const int op = M.lcig[i] & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += M.lcig[i] >> 4; }
                for (uint32_t i = 0; i < n_cig; ++i) { const int op = M.cig[i] & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len_u += M.cig[i] >> 4; }
                const bool mapped = ref_len > 0;
                bool same = n_lcig == n_cig && lifted == ref_pos;
                for (uint32_t i = 0; same && i < n_cig; ++i) same = M.lcig[i] == M.cig[i];
                // both reference windows in LDS with one round trip (the line buffer is free until the lines are rendered)
                const bool win = ref_len <= AFS_LINE / 2 && ref_len_u <= AFS_LINE / 2;
                if (win) {
                    if (mapped) for (uint32_t i = lane; i < (uint32_t)ref_len; i += 64) { const uint64_t a = lifted + i; L.line[i] = (uint8_t)dp_nt4(a < D.n_text ? D.text[a] : 0u); }
                    if (!same) for (uint32_t i = lane; i < (uint32_t)ref_len_u; i += 64) { const uint64_t a = ref_pos + i; L.line[AFS_LINE / 2 + i] = (uint8_t)dp_nt4(a < D.n_text ? D.text[a] : 0u); }
                    __syncthreads();
                }
                uint32_t n_md = 0, dummy = 0;
                int nm = 0;
                if (mapped) nm = afs_md(D, M, M.lcig, n_lcig, lifted, true, n_md, win ? L.line : nullptr);
                const int lift_nm = (same && mapped) ? nm : afs_md(D, M, M.cig, n_cig, ref_pos, false, dummy, win ? L.line + AFS_LINE / 2 : nullptr);
                __syncthreads();
                if (n_md > AFS_MAXMD) { to_host = true; break; }
                f.n_cig = n_cig; f.n_lcig = n_lcig; f.n_md = n_md;
                f.lift_nm = lift_nm; f.lift_pos = ref_pos - LS0.start + 1; f.lift_rname = (int)sid0;
                f.as = R.as[k];
                if (mapped) {
                    const uint32_t lsid = ac_seq_of(P, lifted);
                    f.pos = lifted - P.lift_seqs[lsid].start + 1; f.rname = (int)lsid; f.cigar_star = false; f.rlen = (uint32_t)ref_len; f.nm = nm;
                } else { f.pos = 0; f.rname = -1; f.cigar_star = true; f.rlen = 0; f.unmapped_lft = true; f.nm = 0; f.n_md = 0; }
                f.flag = R.orphan[k] ? 4u : (strand ? 16u : 0u);
                f.zs = R.orphan[k] ? 0 : R.score2_m[k];
                f.mapq = pel_mapq_se(X, f.as, R.score2_m[k], (int32_t)f.rlen, (int32_t)m[k], R.sub_n, unknown);
                const int32_t msc = m[k] <= X.max_len ? X.min_score_of_len[m[k]] : INT32_MAX;
                if (m[k] > X.max_len) unknown = true;
                ok[k] = !f.unmapped_lft && (!R.orphan[k] || f.as >= msc);
            }
            if (!to_host) {
                const uint64_t l1 = m[0], l2 = m[1];
                if (ok[0] && ok[1]) {
                    s[0].pnext = s[1].pos; s[1].pnext = s[0].pos;
                    long long tlen;
                    if (s[1].pos > s[0].pos) { tlen = (long long)((s[1].pos + l2) - s[0].pos); s[0].tlen = tlen; s[1].tlen = -tlen; }
                    else { tlen = (long long)((s[0].pos + l1) - s[1].pos); s[0].tlen = -tlen; s[1].tlen = tlen; }
                    {   // compute_mapq_pe_bwa (mapq.hpp:186-223; pe_host.hpp: mapq_pe_bwa), score_un = 0; size_t / int32_t mixed as in the reference
                        const int32_t sub = R.score2 > 0 ? R.score2 : 0;
                        int32_t mapq = pel_raw_mapq(R.tot - sub, F.smatch);
                        if (R.sub_n > 0) { if ((uint32_t)R.sub_n < X.subn_tab_n) mapq -= X.subn_tab[R.sub_n]; else unknown = true; }
                        if (mapq < 0) mapq = 0;
                        if (mapq > 60) mapq = 60;
                        mapq = (int)__dadd_rn(__dmul_rn((double)mapq, __dsub_rn(1., __dmul_rn(.5, __dadd_rn(0., 0.)))), .499);
                        if (R.tot > 0) {
                            const uint64_t q = (uint64_t)(int64_t)mapq;
                            uint64_t a = s[0].mapq, b = s[1].mapq;
                            a = a > q ? a : (q < a + 40 ? q : a + 40);
                            b = b > q ? b : (q < b + 40 ? q : b + 40);
                            const uint64_t r1 = (uint64_t)(int64_t)pel_raw_mapq(R.mate_score[0] - R.score2_m[0], F.smatch), r2 = (uint64_t)(int64_t)pel_raw_mapq(R.mate_score[1] - R.score2_m[1], F.smatch);
                            a = a < r1 ? a : r1; b = b < r2 ? b : r2;
                            s[0].mapq = a; s[1].mapq = b;
                        }
                    }
                    s[0].as = s[1].as = R.tot;
                    s[0].zs = s[1].zs = R.score2;
                    s[0].flag = s[1].flag = 1 | 2;
                    if (strand) { s[0].flag |= 16 | 64; s[1].flag |= 32 | 128; }
                    else { s[0].flag |= 32 | 64; s[1].flag |= 16 | 128; }
                } else if (ok[0]) {
                    s[0].zs = R.score2_m[0];
                    s[0].flag = 1 | 8 | 64; s[1].flag = 1 | 4 | 128;
                    if (strand) s[0].flag |= 16;
                    s[1].rname = s[0].rname; s[1].pos = s[0].pos; s[1].mapq = s[0].mapq; s[1].cigar_star = true;
                    s[1].pnext = s[0].pnext = s[0].pos;
                    s[1].tlen = s[0].tlen = 0;
                } else if (ok[1]) {
                    s[0].zs = R.score2_m[1];        // sic (aligner_ksw2.hpp:2258)
                    s[0].flag = 1 | 4 | 64; s[1].flag = 1 | 8 | 128;
                    if (!strand) s[1].flag |= 16;
                    s[0].rname = s[1].rname; s[0].pos = s[1].pos; s[0].mapq = s[1].mapq; s[0].cigar_star = true;
                    s[0].pnext = s[1].pnext = s[1].pos;
                    s[0].tlen = s[1].tlen = 0;
                } else {
                    s[0].flag = s[1].flag = 1 | 4 | 8;
                }
            }
        }
        if (unknown) to_host = true;
        // ---- the two lines ----
        uint32_t len2[2] = {0, 0};
        unsigned long long to = 0, words0 = 0;
        for (int k = 0; k < 2 && !to_host; ++k) {
            const pel_fin_t& f = s[k];
            pel_mate_t& M = L.M[k];
            __syncthreads();
            if (lane == 0) {
                uint32_t n = 0, q = 0;
                afs_push(L, n, q, SK_RNAME, (uint32_t)k, nl[k]);
                afs_lits(L, n, q, LT_TAB, 1); afs_num(L, n, q, (int)f.flag); afs_lits(L, n, q, LT_TAB, 1);
                if (f.rname >= 0) afs_push(L, n, q, SK_NAME, (uint32_t)f.rname, PEL_NAME_LEN(f.rname)); else afs_lits(L, n, q, LT_STAR, 1);
                afs_lits(L, n, q, LT_TAB, 1); afs_num(L, n, q, (int)f.pos); afs_lits(L, n, q, LT_TAB, 1); afs_num(L, n, q, (int)f.mapq); afs_lits(L, n, q, LT_TAB, 1);
                if (!f.cigar_star) afs_cigar(L, n, q, M.lcig, f.n_lcig, M.lcig_off, 0u); else afs_lits(L, n, q, LT_STAR, 1);
                afs_lits(L, n, q, LT_TAB, 1);
                if (same_name) afs_lits(L, n, q, LT_OPS + 7, 1); else afs_push(L, n, q, SK_RNAME, (uint32_t)(1 - k), nl[1 - k]);
                afs_lits(L, n, q, LT_TAB, 1); afs_num(L, n, q, (int)f.pnext); afs_lits(L, n, q, LT_TAB, 1); afs_num(L, n, q, (int)(uint64_t)f.tlen); afs_lits(L, n, q, LT_TAB, 1);
                L.seq_at = q; afs_push(L, n, q, SK_SEQ, 0, m[k]); afs_lits(L, n, q, LT_TAB, 1);
                L.qual_at = F.quals ? q : ~0u;
                if (F.quals) afs_push(L, n, q, SK_QUAL, 0, m[k]); else afs_lits(L, n, q, LT_STAR, 1);
                if (!(f.flag & 4) || f.unmapped_lft) {
                    afs_lits(L, n, q, LT_AS, 6); afs_num(L, n, q, f.as); afs_lits(L, n, q, LT_NM, 6); afs_num(L, n, q, f.nm);
                    if (f.zs != 0) { afs_lits(L, n, q, LT_ZS, 6); afs_num(L, n, q, f.zs); }
                    afs_lits(L, n, q, LT_MD, 6);
                    {
                        uint32_t w = 0;
                        for (uint32_t i = 0; i < f.n_md; ++i) {
                            const uint32_t it = M.md_item[i], ty = it & 3u;
                            M.md_off[i] = (uint16_t)w;
                            w += afs_ndig((it >> 2) & 0x3FFu) + (ty == 1 ? 1u : ty == 2 ? 1u + ((it >> 12) & 0x1FFu) : 0u);
                        }
                        M.md_off[f.n_md] = (uint16_t)w;
                        afs_push(L, n, q, SK_MD, f.n_md, w);
                    }
                    afs_lits(L, n, q, LT_OA, 6);
                    if (f.lift_rname >= 0) afs_push(L, n, q, SK_NAME, (uint32_t)f.lift_rname, PEL_NAME_LEN(f.lift_rname)); else afs_lits(L, n, q, LT_STAR, 1);
                    afs_lits(L, n, q, LT_COMMA, 1); afs_num(L, n, q, (int)f.lift_pos); afs_lits(L, n, q, (f.flag & 16) ? LT_MINUS : LT_PLUS, 3);
                    if (f.filled) afs_cigar(L, n, q, M.cig, f.n_cig, M.cig_off, 1u); else afs_lits(L, n, q, LT_STAR, 1);
                    afs_lits(L, n, q, LT_COMMA, 1); afs_num(L, n, q, (int)f.mapq); afs_lits(L, n, q, LT_COMMA, 1); afs_num(L, n, q, f.lift_nm); afs_lits(L, n, q, LT_SEMI, 1);
                    afs_lits(L, n, q, LT_AA, 6);
                    if (f.filled) for (uint32_t i = 0; i < R.n_alt[k]; ++i) {
                        const uint32_t s2 = M.alt_sid[i];
                        afs_push(L, n, q, SK_NAME, s2, PEL_NAME_LEN(s2));
                        afs_lits(L, n, q, LT_COMMA, 1); afs_num(L, n, q, (int)M.alt_p1[i]); afs_lits(L, n, q, LT_COMMA, 1); afs_num(L, n, q, M.alt_score[i]); afs_lits(L, n, q, LT_SEMI, 1);
                    }
                }
                afs_lits(L, n, q, LT_NL, 1);
                if (n <= AFS_MAXSEG) L.seg_off[n] = (uint16_t)(q < 0xFFFFu ? q : 0xFFFFu);
                L.n_seg = n; L.total = q;
            }
            __syncthreads();
            const uint32_t n_seg = L.n_seg, p = L.total;
            if (n_seg > AFS_MAXSEG || p > AFS_LINE) { to_host = true; break; }
            const uint32_t s_at = L.seq_at, q_at = L.qual_at, mk = m[k], holes = mk + (q_at != ~0u ? mk : 0u);
            for (uint32_t i = lane; i < mk; i += 64) L.line[s_at + i] = M.seq[i];
            if (q_at != ~0u) for (uint32_t i = lane; i < mk; i += 64) L.line[q_at + i] = F.quals[rev[k] ? off[k] + mk - 1 - i : off[k] + i];
            for (uint32_t i = lane; i + holes < p; i += 64) {
                uint32_t b = i;
                if (b >= s_at) b += mk;
                if (q_at != ~0u && b >= q_at) b += mk;
                uint32_t lo = 0, hi = n_seg;
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)L.seg_off[mid] <= b) lo = mid; else hi = mid; }
                const uint32_t d = b - L.seg_off[lo], kind = L.seg_kind[lo], val = L.seg_val[lo];
                uint8_t ch;
                if (kind == SK_LIT) ch = (uint8_t)afs_lit[val + d];
                else if (kind == SK_NUM || kind == SK_NEG) {
                    const uint32_t len = (uint32_t)L.seg_off[lo + 1] - L.seg_off[lo];
                    if (kind == SK_NEG && d == 0) ch = '-';
                    else { uint32_t u = val; for (uint32_t t = d + 1; t < len; ++t) u /= 10u; ch = (uint8_t)('0' + u % 10u); }
                }
                else if (kind == SK_NAME) ch = names_lds ? L.names[L.name_off[val] + d] : F.snames[F.sname_off[val] + d];
                else if (kind == SK_RNAME) ch = F.rnames[nb[val] + d];
                else if (kind == SK_CIG) {
                    const uint16_t* offs = (val & 1u) ? M.cig_off : M.lcig_off; const uint32_t* cg = (val & 1u) ? M.cig : M.lcig;
                    uint32_t a = 0, z = val >> 1;
                    while (z - a > 1) { const uint32_t mid = (a + z) >> 1; if ((uint32_t)offs[mid] <= d) a = mid; else z = mid; }
                    const uint32_t e = d - offs[a], nd = (uint32_t)offs[a + 1] - offs[a] - 1u;
                    if (e == nd) ch = (uint8_t)afs_lit[LT_OPS + (cg[a] & 0xfu)];
                    else { uint32_t u = cg[a] >> 4; for (uint32_t t = e + 1; t < nd; ++t) u /= 10u; ch = (uint8_t)('0' + u % 10u); }
                } else {                            // SK_MD
                    uint32_t a = 0, z = val;
                    while (z - a > 1) { const uint32_t mid = (a + z) >> 1; if ((uint32_t)M.md_off[mid] <= d) a = mid; else z = mid; }
                    const uint32_t it = M.md_item[a], ty = it & 3u, run = (it >> 2) & 0x3FFu, e = d - M.md_off[a], nd = afs_ndig(run);
                    if (e < nd) { uint32_t u = run; for (uint32_t t = e + 1; t < nd; ++t) u /= 10u; ch = (uint8_t)('0' + u % 10u); }
                    else if (ty == 1) { const uint32_t bc = (it >> 12) & 7u; ch = (uint8_t)afs_lit[LT_BASES + (bc > 4 ? 4 : bc)]; }
                    else if (e == nd) ch = '^';
                    else { const uint64_t ta = M.lifted + (it >> 21) + (e - nd - 1u); ch = (uint8_t)afs_lit[LT_BASES + dp_nt4(ta < D.n_text ? D.text[ta] : 0u)]; }
                }
                L.line[b] = ch;
            }
            __syncthreads();
            // out: the pair's own place in the pool (room for two lines of the staging's size), each line stored as it is rendered
            const unsigned long long words = (unsigned long long)((p + 7) >> 3);
            if (k == 0) { to = (unsigned long long)r_in * PEL_PAIR_WORDS; words0 = words; }
            const unsigned long long at = k == 0 ? to : to + words0;
            const uint64_t* src = reinterpret_cast<const uint64_t*>(L.line);
            for (unsigned long long w = lane; w < words; w += 64) X.txt_pool[at + w] = src[w];
            len2[k] = p;
        }
        if (lane == 0) {
            if (to_host) { X.dev_len[2 * r_in] = 0; X.dev_len[2 * r_in + 1] = 0; X.dev_off[2 * r_in] = 0; X.dev_off[2 * r_in + 1] = 0; atomicAdd(&X.dev_sum[0], 1ull); }
            else {
                X.dev_len[2 * r_in] = len2[0]; X.dev_len[2 * r_in + 1] = len2[1]; X.dev_off[2 * r_in] = to; X.dev_off[2 * r_in + 1] = to + words0;
                // al.aligned: tot >= min_score (aligner_ksw2.hpp:1000-1010, 1326)
                if (finalized && R.tot >= X.min_score_of_len[m[0]] + X.min_score_of_len[m[1]]) atomicAdd(&X.dev_sum[8 + 8 * (blockIdx.x % 16)], 1ull);
            }
        }
    }
#undef PEL_NAME_LEN
}
