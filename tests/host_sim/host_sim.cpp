// TEST HARNESS ONLY — not part of the product, never linked into libmoni_hip.so.
// Replays the per-lane functions of moni_align_amd/csrc/seed_core.h (the code the HIP kernels run)
// sequentially on the host over the host copy of the index image, with the same orchestration as
// moni_seed_run, so that the device layout and step logic can be checked against the oracle on a
// machine without a GPU.  The GPU tests (-m gpu) check the real kernels through the C ABI.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../moni_align_amd/csrc/image.hpp"
#include "../../moni_align_amd/csrc/lift_build.hpp"
#include "../../moni_align_amd/csrc/seed_core.h"
#include "../../moni_align_amd/csrc/align_host.hpp"
#include "../../moni_align_amd/csrc/align_core.h"
#include "../../moni_align_amd/csrc/pe_core.h"
#include "../../moni_align_amd/csrc/pe_host.hpp"
#include "../../moni_align_amd/csrc/pe_big.h"
#include "../../oracle/ksw2.hpp"      // CPU stand-in for extz_kernel in this harness (tests may use the oracle)
#include "../../oracle/align_pe.hpp"  // ... and for the local-alignment requests of orphan recovery (klib ksw_align as restated there)

struct Sim {
    HostImage img;
    std::vector<uint8_t> text;
    std::vector<uint32_t> name_id;
    lds_tables_t L;
    // last result
    std::vector<uint64_t> ptr;
    std::vector<moni_u64x2> blk;      // workspace layout of the last batch (seed_core.h)
    std::vector<moni_mem_t> mems;
    std::vector<uint64_t> occs, read_mem_off;
    std::vector<uint32_t> aux;      // per MEM slot: plain / has halves / is a half (seed_core.h)
    uint64_t counters[4];
    uint64_t max_len = 0, n_reads = 0;
    mh::HostIndex hix;
    std::vector<uint64_t> pdir;
};

extern "C" {

void* sim_create(const moni_flat_index_t* f) {
    Sim* S = new Sim();
    if (S->img.build(*f)) { fprintf(stderr, "host_sim: %s\n", S->img.err.c_str()); delete S; return nullptr; }
    if (getenv("MH_TIMES")) { uint64_t ok = 0, cov = 0; for (uint64_t k = 0; k < f->r; ++k) if ((S->img.frows[k].w[0] >> 58) & 1) { ++ok; cov += f->starts[k + 1] - f->starts[k]; }
        fprintf(stderr, "fast rows: %.4f of runs, %.4f of BWT positions\n", (double)ok / f->r, (double)cov / f->n); }
    S->text.assign(f->text, f->text + f->n - 1);
    S->text.resize(S->text.size() + 16, 0);
    S->name_id.resize(f->n_seq);
    for (uint64_t i = 0; i < f->n_seq; ++i) S->name_id[i] = (uint32_t)i;
    memcpy(S->L.code, S->img.T.code, 256);
    memcpy(S->L.compl_tab, S->img.T.compl_tab, 256);
    for (int i = 0; i < 256; ++i) S->L.c2[i] = base_acgt((uint32_t)i) ? (uint8_t)base2((uint32_t)i) : (uint8_t)4;
    memcpy(S->L.abs_run, S->img.T.abs_run, sizeof(S->L.abs_run));
    memcpy(S->L.abs_pos, S->img.T.abs_pos, sizeof(S->L.abs_pos));
    for (int i = 0; i < MONI_MAX_SIGMA; ++i) { S->L.rec_base[i] = S->img.K.rec_base[i]; S->L.rec_cnt[i] = S->img.K.rec_cnt[i]; S->L.hot_slot[i] = S->img.K.hot_slot[i]; }
    S->hix.n_text = f->n - 1; S->hix.w = f->w; S->hix.text = S->text.data();
    S->hix.seq_starts.assign(f->seq_starts, f->seq_starts + f->n_seq + 1);
    const char* p = f->seq_names;
    for (uint64_t i = 0; i < f->n_seq; ++i) { std::string nm = p ? std::string(p) : ("seq" + std::to_string(i)); if (p) p += nm.size() + 1; S->hix.names.push_back(nm); }
    { LiftTables LT; std::string err; if (LT.build(*f, err)) { fprintf(stderr, "host_sim: %s\n", err.c_str()); delete S; return nullptr; }
      S->hix.lift_seqs = LT.seqs; S->hix.lift_runs = LT.runs; S->pdir = LT.pdir; }
    return S;
}
void sim_destroy(void* s) { delete (Sim*)s; }

int sim_seed_run(void* s, const uint8_t* seq, const uint64_t* offs, uint64_t n_reads, const moni_seed_params_t* prm,
                 uint32_t tmp_cap, uint32_t pool_rows) {
    Sim* S = (Sim*)s;
    const moni_consts_t& K = S->img.K;
    const uint64_t n_tasks = 2 * n_reads;
    uint64_t mx = 0;
    for (uint64_t i = 0; i < n_reads; ++i) mx = std::max<uint64_t>(mx, offs[i + 1] - offs[i]);
    S->max_len = mx; S->n_reads = n_reads;
    // the workspace layout of reads_upload (moni_hip.hip): per block of 32 reads as many steps as its longest read has
    const uint64_t n_blk = (n_reads + 31) / 32;
    S->blk.assign(n_blk + 1, moni_u64x2());
    {
        uint64_t pw = 0, qw = 0;
        for (uint64_t k = 0; k < n_blk; ++k) {
            uint64_t lb = 0;
            for (uint64_t i = 32 * k; i < n_reads && i < 32 * k + 32; ++i) lb = std::max<uint64_t>(lb, offs[i + 1] - offs[i]);
            S->blk[k].x = qw; S->blk[k].y = pw;
            qw += 64 * lb; pw += 64 * ws_pat_words(lb);
        }
        S->blk[n_blk].x = qw; S->blk[n_blk].y = pw;
    }
    const moni_u64x2* blk = S->blk.data();
    S->ptr.assign(S->blk[n_blk].x + 1, 0);
    unsigned long long cnt[4] = {0, 0, 0, 0};
    std::vector<uint64_t> pat(S->blk[n_blk].y + 1);
    // pack_task reads aligned 8-byte words: the device buffer is aligned and padded by 16 bytes, so is this copy
    std::vector<uint64_t> seq_pad((offs[n_reads] + 16 + 7) / 8 + 1, 0);
    memcpy(seq_pad.data(), seq, offs[n_reads]);
    seq = reinterpret_cast<const uint8_t*>(seq_pad.data());
    for (uint64_t t = 0; t < n_tasks; ++t) pack_task(S->L, seq, offs, blk, t, pat.data());
    for (uint64_t t = 0; t < n_tasks; t += 2)
        ms_task<2>(K, S->L, S->img.rows.data(), S->img.frows.data(), S->img.cr.data(), S->img.recs.data(), pat.data(), offs, blk, n_tasks, t, S->ptr.data(), cnt[0], cnt[1]);
    std::vector<uint32_t> cnt_m(n_tasks + 1), cnt_s(n_tasks + 1);
    std::vector<moni_u64x2> slots(n_tasks * MONI_MEM_SLOTS + 1);
    const uint32_t split_on = prm->report_mems ? 0 : 1;
    // the 2-bit text and its exception bitmap as the index builds them on the device; MONI_MEM_BYTES: every comparison byte by byte
    const uint64_t n_text = K.n_text;
    std::vector<uint64_t> text2(n_text / 32 + 2, 0), pw(64, 0);
    mem_fast_t F;
    F.exc_sh = text2_exc_shift(n_text, MONI_EXC_BITS);
    std::vector<uint32_t> exc((((n_text >> F.exc_sh) + 1) + 31) / 32 + 1, 0);
    for (uint64_t w = 0; w < text2.size(); ++w) { bool bad; text2[w] = text2_word(S->text.data(), n_text, w, bad); if (bad) { const uint64_t b = (32 * w) >> F.exc_sh; exc[b >> 5] |= 1u << (b & 31u); } }
    F.text2 = getenv("MONI_MEM_BYTES") ? nullptr : text2.data(); F.exc = exc.data(); F.pw = pw.data(); F.pw_stride = 1; F.pw_words = getenv("MONI_SIM_PW") ? (uint32_t)atoi(getenv("MONI_SIM_PW")) : 8u;
    for (uint64_t t = 0; t < n_tasks; ++t)
        mem_task<false>(K, F, S->text.data(), pat.data(), offs, blk, t, S->ptr.data(), prm->min_len, split_on, cnt_m.data(), cnt_s.data(),
                        nullptr, nullptr, nullptr, slots.data(), cnt[3]);
    S->read_mem_off.assign(n_reads + 1, 0);
    for (uint64_t r = 0; r < n_reads; ++r)
        S->read_mem_off[r + 1] = S->read_mem_off[r] + cnt_m[2 * r] + cnt_m[2 * r + 1] + 2ull * (cnt_s[2 * r] + cnt_s[2 * r + 1]);
    const uint64_t n_mems = S->read_mem_off[n_reads];
    S->mems.assign(n_mems + 1, moni_mem_t());
    std::vector<uint32_t> aux(n_mems + 1);
    unsigned long long dummy = 0;
    for (uint64_t t = 0; t < n_tasks; ++t)
        mem_task<true>(K, F, S->text.data(), pat.data(), offs, blk, t, S->ptr.data(), prm->min_len, split_on, cnt_m.data(), cnt_s.data(),
                       S->read_mem_off.data(), S->mems.data(), aux.data(), slots.data(), dummy);
    std::vector<uint64_t> tmp(n_mems * tmp_cap + 1), lowers(n_mems + 1);
    std::vector<uint32_t> pool((size_t)pool_rows * K.n_seq + 1);
    uint32_t small[2] = {0, 0};
    occ_args_t A;
    A.text = S->text.data();
    A.phi.recs = S->img.phi.data(); A.phi.dir = S->img.phi_dir.data();
    A.phi_inv.recs = S->img.phi_inv.data(); A.phi_inv.dir = S->img.phi_inv_dir.data();
    A.seq_starts = S->img.seq_starts.data(); A.name_id = S->name_id.data(); A.mems = S->mems.data(); A.aux = aux.data();
    A.read_mem_off = S->read_mem_off.data(); A.n_mems = n_mems; A.occs = nullptr; A.tmp = tmp.data(); A.lowers = lowers.data();
    A.tmp_cap = tmp_cap; A.filter_seeds = prm->filter_seeds; A.n_seeds_thr = prm->n_seeds_thr; A.pool_rows = pool_rows;
    A.pool = pool.data(); A.pool_next = &small[0]; A.error_flag = &small[1]; A.counters = nullptr;
    for (uint64_t g = 0; g < n_mems; ++g) occ_task<false>(K, A, g, cnt[2]);
    if (small[1]) return MONI_ENOMEM;
    uint64_t acc = 0;
    for (uint64_t g = 0; g < n_mems; ++g) { S->mems[g].occ_off = acc; acc += S->mems[g].occ_cnt; }
    S->occs.assign(acc + 1, 0);
    A.occs = S->occs.data();
    small[0] = small[1] = 0;
    for (uint64_t g = 0; g < n_mems; ++g) occ_task<true>(K, A, g, dummy);
    if (small[1]) return MONI_ENOMEM;
    S->occs.resize(acc);
    S->mems.resize(n_mems);
    S->aux = aux;
    for (int i = 0; i < 4; ++i) S->counters[i] = cnt[i];
    return MONI_OK;
}

uint64_t sim_n_mems(void* s) { return ((Sim*)s)->mems.size(); }
uint64_t sim_n_occs(void* s) { return ((Sim*)s)->occs.size(); }
void sim_fetch(void* s, moni_mem_t* mems, uint64_t* occs, uint64_t* read_mem_off, uint64_t* counters) {
    Sim* S = (Sim*)s;
    if (!S->mems.empty()) memcpy(mems, S->mems.data(), S->mems.size() * sizeof(moni_mem_t));
    if (!S->occs.empty()) memcpy(occs, S->occs.data(), S->occs.size() * 8);
    memcpy(read_mem_off, S->read_mem_off.data(), S->read_mem_off.size() * 8);
    memcpy(counters, S->counters, 32);
}
// pointers in the layout of moni_ms_query_batch
void sim_fetch_pointers(void* s, const uint64_t* offs, uint64_t* pointers) {
    Sim* S = (Sim*)s;
    for (uint64_t rd = 0; rd < S->n_reads; ++rd) {
        const uint64_t off = offs[rd] - offs[0], m = offs[rd + 1] - offs[rd];
        for (uint64_t st = 0; st < 2; ++st)
            for (uint64_t k = 0; k < m; ++k) pointers[2 * off + st * m + k] = S->ptr[ws_ptr_base(S->blk.data(), 2 * rd + st) + (m - 1 - k) * 64];
    }
}
void sim_phi(void* s, uint64_t i, int inverse, uint64_t* out2) {
    Sim* S = (Sim*)s;
    phi_tab_t P;
    P.recs = inverse ? S->img.phi_inv.data() : S->img.phi.data();
    P.dir = inverse ? S->img.phi_inv_dir.data() : S->img.phi_dir.data();
    phi_step(P, S->img.K, i, out2[0], out2[1]);
}

// ---- the host pipeline (align_host.hpp) driven by CPU stand-ins: replayed seed kernels + the oracle's ksw2 ----
struct SimBackend : mh::Backend {
    Sim* S; const uint8_t* seq; const uint64_t* offs; uint64_t n_reads;
    int seed(const moni_seed_params_t& p, std::vector<moni_mem_t>& mems, std::vector<uint64_t>& occs, std::vector<uint64_t>& rmo) override {
        int rc = sim_seed_run(S, seq, offs, n_reads, &p, 16, 4096);
        if (rc) return rc;
        mems = S->mems; occs = S->occs; rmo = S->read_mem_off;
        return 0;
    }
    int dp(const moni_dp_params_t& p, const std::vector<moni_dp_task_t>& tasks, std::vector<moni_dp_result_t>& res, std::vector<uint32_t>& cig) override {
        res.resize(tasks.size());
        cig.clear();
        for (size_t i = 0; i < tasks.size(); ++i) {
            const moni_dp_task_t& t = tasks[i];
            std::vector<uint8_t> q(t.qlen > 0 ? t.qlen : 0), tg(t.tlen > 0 ? t.tlen : 0);
            for (int k = 0; k < t.qlen; ++k) {
                uint8_t c = mh::nt4_of(seq[(t.reserved & DP_Q_REV) ? t.q_off - k : t.q_off + k]);
                if ((t.reserved & DP_Q_COMP) && c < 4) c = 3 - c;
                q[k] = c;
            }
            for (int k = 0; k < t.tlen; ++k) {
                const uint64_t a = (t.reserved & DP_T_REV) ? t.t_off - k : t.t_off + k;
                tg[k] = mh::nt4_of(a < S->hix.n_text ? S->text[a] : 0);
            }
            if (t.flag & DP_EZ_LOCAL) {
                const oracle::aligner_pe::kswr_t k = oracle::aligner_pe::ksw_align(t.qlen, q.data(), t.tlen, tg.data(), p.m, p.mat, p.q, p.e);
                moni_dp_result_t& r = res[i];
                memset(&r, 0, sizeof r);
                r.score = k.score; r.max_t = k.te; r.max_q = k.qe; r.mte = k.tb; r.mte_q = k.qb; r.cigar_off = (uint32_t)cig.size();
                continue;
            }
            oracle::ksw_extz_t ez;
            memset(&ez, 0, sizeof ez);
            oracle::ksw_extz2_restated(t.qlen, q.data(), t.tlen, tg.data(), p.m, p.mat, p.q, p.e, p.w, p.zdrop, p.end_bonus, t.flag, &ez);
            moni_dp_result_t& r = res[i];
            r.max = (int32_t)ez.max; r.max_q = ez.max_q; r.max_t = ez.max_t; r.mqe = ez.mqe; r.mqe_t = ez.mqe_t; r.mte = ez.mte; r.mte_q = ez.mte_q;
            r.score = ez.score; r.reach_end = ez.reach_end; r.zdropped = ez.zdropped; r.n_cigar = ez.n_cigar; r.cigar_off = (uint32_t)cig.size();
            for (int k = 0; k < ez.n_cigar; ++k) cig.push_back(ez.cigar[k]);
            free(ez.cigar);
        }
        return 0;
    }
};

char* sim_align_batch(void* s, const uint8_t* seq, const uint64_t* offs, uint64_t n_reads, const uint8_t* names, const uint64_t* name_off,
                      const uint8_t* quals, int threads, uint64_t* out_len, uint64_t* stats5) {
    Sim* S = (Sim*)s;
    SimBackend be;
    be.S = S; be.seq = seq; be.offs = offs; be.n_reads = n_reads;
    moni_align_params_t P;
    memset(&P, 0, sizeof P);
    P.min_len = 25; P.ext_len = 100; P.check_k = 5; P.region_dist = 10; P.filter_seeds = 1; P.n_seeds_thr = 1000; P.filter_freq = 1;
    P.left_mem_check = 1; P.freq_thr = 0.5; P.smatch = 2; P.smismatch = 4; P.gapo = 4; P.gapo2 = 13; P.gape = 2; P.gape2 = 1;
    P.end_bonus = 400; P.w = -1; P.zdrop = -1; P.max_dist_x = 500; P.max_dist_y = 100; P.max_iter = 10; P.max_pred = 5;
    P.min_chain_score = 40; P.min_chain_length = 1; P.host_threads = threads;
    std::string out;
    mh::AlignStats st;
    int rc = mh::align_batch(be, S->hix, P, seq, offs, n_reads, names, name_off, quals, out, st);
    if (rc) return nullptr;
    char* buf = (char*)malloc(out.size() + 1);
    memcpy(buf, out.data(), out.size() + 1);
    *out_len = out.size();
    if (stats5) { stats5[0] = st.reads; stats5[1] = st.aligned; stats5[2] = st.dp_tasks; stats5[3] = st.dp_cells; stats5[4] = st.dp_rounds; }
    if (getenv("MH_TIMES")) fprintf(stderr, "host_sim times: seed %.3f chain %.3f dp %.3f host %.3f s\n", st.t_seed, st.t_chain, st.t_dp, st.t_host);
    return buf;
}
// ---- the device-side per-read logic (align_core.h) replayed on the host: what the align kernel's lane 0 runs ----
char* sim_align_core_batch(void* s, const uint8_t* seq, const uint64_t* offs, uint64_t n_reads, const uint8_t* names, const uint64_t* name_off,
                           const uint8_t* quals, uint64_t* out_len, uint64_t* stats5) {
    Sim* S = (Sim*)s;
    SimBackend be;
    be.S = S; be.seq = seq; be.offs = offs; be.n_reads = n_reads;
    moni_align_params_t P;
    memset(&P, 0, sizeof P);
    P.min_len = 25; P.ext_len = 100; P.check_k = 5; P.region_dist = 10; P.filter_seeds = 1; P.n_seeds_thr = 1000; P.filter_freq = 1;
    P.left_mem_check = 1; P.freq_thr = 0.5; P.smatch = 2; P.smismatch = 4; P.gapo = 4; P.gapo2 = 13; P.gape = 2; P.gape2 = 1;
    P.end_bonus = 400; P.w = -1; P.zdrop = -1; P.max_dist_x = 500; P.max_dist_y = 100; P.max_iter = 10; P.max_pred = 5;
    P.min_chain_score = 40; P.min_chain_length = 1; P.host_threads = 1;
    moni_seed_params_t sp{P.min_len, P.filter_seeds, P.n_seeds_thr, 0};
    std::vector<moni_mem_t> gm; std::vector<uint64_t> go, rmo;
    if (be.seed(sp, gm, go, rmo)) return nullptr;
    ac_params_t AP;
    AP.min_len = P.min_len; AP.ext_len = P.ext_len; AP.check_k = P.check_k; AP.region_dist = P.region_dist; AP.filter_freq = P.filter_freq;
    AP.left_mem_check = P.left_mem_check; AP.freq_thr = P.freq_thr; AP.smatch = P.smatch; AP.gapo = P.gapo; AP.gapo2 = P.gapo2; AP.gape = P.gape;
    AP.gape2 = P.gape2; AP.max_dist_x = P.max_dist_x; AP.max_dist_y = P.max_dist_y; AP.max_iter = P.max_iter; AP.max_pred = P.max_pred;
    AP.min_chain_score = P.min_chain_score; AP.min_chain_length = P.min_chain_length; AP.n_text = S->hix.n_text; AP.n_seq = (uint32_t)S->hix.names.size();
    AP.seq_starts = S->hix.seq_starts.data();
    AP.lift_seqs = S->hix.lift_seqs.data(); AP.lift_runs = S->hix.lift_runs.data(); AP.pdir = S->pdir.data();
    moni_dp_params_t dp;
    memset(&dp, 0, sizeof dp);
    dp.m = 5;
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) dp.mat[i * 5 + j] = i == j ? P.smatch : (int8_t)-P.smismatch; }
    dp.q = P.gapo; dp.e = P.gape; dp.w = -1; dp.zdrop = -1; dp.end_bonus = P.end_bonus;
    mh::Aligner A(S->hix, P, seq, offs);
    std::string out, name, sq, ql;
    uint64_t n_aligned = 0, n_over = 0, n_tasks = 0, n_rounds = 0;
    ac_ws_t* W = new ac_ws_t();
    mh::Aligner::OutBuf emit_ob; std::vector<char> emit_md; uint64_t n_emit_diff = 0;
    for (uint64_t r = 0; r < n_reads; ++r) {
        W->off = offs[r] - offs[0]; W->m = (uint32_t)(offs[r + 1] - offs[r]);
        W->min_score = (int32_t)(20 + 8 * log((double)W->m));
        std::vector<moni_dp_result_t> res;
        std::vector<uint32_t> cig;
        if (ac_init(*W, AP, gm.data(), rmo[r], rmo[r + 1], go.data())) {
            ac_drive(*W, AP, nullptr, nullptr);
            while (!W->overflow && W->stage != AC_DONE) {
                std::vector<moni_dp_task_t> tasks(W->tasks, W->tasks + W->n_tasks);
                n_tasks += tasks.size(); ++n_rounds;
                if (be.dp(dp, tasks, res, cig)) return nullptr;
                ac_drive(*W, AP, res.data(), cig.data());
            }
        }
        mh::Sam Sm;
        if (W->overflow) ++n_over;
        else if (W->aligned) {
            A.finish_record(W->m, W->off, W->fill.strand, W->fill.ref_pos, W->fill.score, W->score2, W->cigar, W->n_cigar, W->alt_pos, W->alt_score, W->n_alt, Sm);
            ++n_aligned;
        }
        const uint8_t* sp0 = seq + W->off;
        name.assign((const char*)names + name_off[r], (const char*)names + name_off[r + 1]);
        sq.resize(W->m);
        if (Sm.rev_read) for (uint32_t k = 0; k < W->m; ++k) sq[k] = (char)mh::compl_of(sp0[W->m - 1 - k]); else sq.assign((const char*)sp0, (const char*)sp0 + W->m);
        if (quals) { const uint8_t* qv = quals + W->off; ql.resize(W->m); if (Sm.rev_read) for (uint32_t k = 0; k < W->m; ++k) ql[k] = (char)qv[W->m - 1 - k]; else ql.assign((const char*)qv, (const char*)qv + W->m); }
        if (!W->aligned) Sm.flag = 4;
        const size_t line_at = out.size();
        mh::Aligner::sam_write(out, Sm, name, sq, quals ? &ql : nullptr);
        // the host stage's one-pass emitter (what moni_align_batch runs behind align_kernel) must spell the same record, with MD/NM
        // computed here and with MD/NM handed in (as the kernel does)
        if (!W->overflow) {
            mh::moni_alt_like alts[AC_MAX_ALT];
            for (uint32_t k = 0; k < W->n_alt; ++k) { alts[k].pos = W->alt_pos[k]; alts[k].score = W->alt_score[k]; alts[k].pad = 0; }
            const bool al = W->aligned != 0;
            emit_ob.len = 0;
            if (!A.emit_record(emit_ob, emit_md, name.data(), name.size(), seq + W->off, quals ? quals + W->off : nullptr, W->m, al, W->fill.strand, W->fill.ref_pos,
                               W->fill.score, W->score2, W->cigar, al ? W->n_cigar : 0, alts, al ? W->n_alt : 0)) return nullptr;
            if (emit_ob.len != out.size() - line_at || memcmp(emit_ob.base, out.data() + line_at, emit_ob.len) != 0) ++n_emit_diff;
            if (al) {
                const std::string mdz = Sm.md;
                emit_ob.len = 0;
                if (!A.emit_record(emit_ob, emit_md, name.data(), name.size(), seq + W->off, quals ? quals + W->off : nullptr, W->m, al, W->fill.strand, W->fill.ref_pos,
                                   W->fill.score, W->score2, W->cigar, W->n_cigar, alts, W->n_alt, mdz.data(), (uint32_t)mdz.size(), (int)Sm.nm, (int)Sm.lift_nm)) return nullptr;
                if (emit_ob.len != out.size() - line_at || memcmp(emit_ob.base, out.data() + line_at, emit_ob.len) != 0) ++n_emit_diff;
            }
        }
    }
    emit_ob.release();
    if (n_emit_diff) { delete W; return nullptr; }
    delete W;
    char* buf = (char*)malloc(out.size() + 1);
    memcpy(buf, out.data(), out.size() + 1);
    *out_len = out.size();
    if (stats5) { stats5[0] = n_reads; stats5[1] = n_aligned; stats5[2] = n_tasks; stats5[3] = n_over; stats5[4] = n_rounds; }
    return buf;
}
// ---- the paired-end per-pair logic (pe_core.h) replayed on the host + the host finishing (pe_host.hpp).  Reads 2p / 2p + 1 are the mates
// of pair p.  finalize == 0: the learn pass (learn[4*p .. 4*p+4) = aligned, best tot, second tot, dist; min_score in learn_min[p]) ----
char* sim_align_pe_batch(void* s, const uint8_t* seq, const uint64_t* offs, uint64_t n_pairs, const uint8_t* names, const uint64_t* name_off,
                         const uint8_t* quals, int finalize, double mean, double std_dev, int find_orphan, long long* learn, uint64_t* out_len, uint64_t* stats5) {
    Sim* S = (Sim*)s;
    SimBackend be;
    const uint64_t n_reads = 2 * n_pairs;
    be.S = S; be.seq = seq; be.offs = offs; be.n_reads = n_reads;
    moni_align_params_t P;
    memset(&P, 0, sizeof P);
    P.min_len = 25; P.ext_len = 100; P.check_k = 5; P.region_dist = 10; P.filter_seeds = 1; P.n_seeds_thr = 1000; P.filter_freq = 1;
    P.left_mem_check = 1; P.freq_thr = 0.5; P.smatch = 2; P.smismatch = 4; P.gapo = 4; P.gapo2 = 13; P.gape = 2; P.gape2 = 1;
    P.end_bonus = 400; P.w = -1; P.zdrop = -1; P.max_dist_x = 500; P.max_dist_y = 100; P.max_iter = 10; P.max_pred = 5;
    P.min_chain_score = 40; P.min_chain_length = 1; P.host_threads = 1;
    moni_seed_params_t sp{P.min_len, P.filter_seeds, P.n_seeds_thr, 0};
    std::vector<moni_mem_t> gm; std::vector<uint64_t> go, rmo;
    if (be.seed(sp, gm, go, rmo)) return nullptr;
    const std::vector<uint32_t>& aux = S->aux;
    pe_params_t PP;
    PP.pen_tab = nullptr; PP.pen_tab_n = 0; PP.secondary_chains = 0;
    ac_params_t& AP = PP.P;
    AP.min_len = P.min_len; AP.ext_len = P.ext_len; AP.check_k = P.check_k; AP.region_dist = P.region_dist; AP.filter_freq = P.filter_freq;
    AP.left_mem_check = P.left_mem_check; AP.freq_thr = P.freq_thr; AP.smatch = P.smatch; AP.gapo = P.gapo; AP.gapo2 = P.gapo2; AP.gape = P.gape;
    AP.gape2 = P.gape2; AP.max_dist_x = P.max_dist_x; AP.max_dist_y = P.max_dist_y; AP.max_iter = P.max_iter; AP.max_pred = P.max_pred;
    AP.min_chain_score = P.min_chain_score; AP.min_chain_length = P.min_chain_length; AP.n_text = S->hix.n_text; AP.n_seq = (uint32_t)S->hix.names.size();
    AP.seq_starts = S->hix.seq_starts.data();
    AP.lift_seqs = S->hix.lift_seqs.data(); AP.lift_runs = S->hix.lift_runs.data(); AP.pdir = S->pdir.data();
    PP.smismatch = P.smismatch; PP.max_penalty = std::max(P.smatch + P.smismatch, P.gapo + P.gape); PP.filter_dir = 1; PP.finalize = finalize ? 1 : 0;
    PP.dir_thr = 50.0; PP.mean = (float)mean; PP.std_dev = (float)std_dev;
    PP.find_orphan = (find_orphan & 1) ? 1 : 0; PP.secondary_chains = (find_orphan & 2) ? 1 : 0; PP.w = (uint32_t)S->hix.w; PP.ins_mean = mean; PP.ins_std_dev = std_dev;      // bit 1: -Z
    moni_dp_params_t dp;
    memset(&dp, 0, sizeof dp);
    dp.m = 5;
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) dp.mat[i * 5 + j] = i == j ? P.smatch : (int8_t)-P.smismatch; }
    dp.q = P.gapo; dp.e = P.gape; dp.w = -1; dp.zdrop = -1; dp.end_bonus = P.end_bonus;
    mh::Aligner A(S->hix, P, seq, offs);
    std::string out;
    uint64_t n_aligned = 0, n_over = 0, n_tasks = 0, n_rounds = 0;
    pe_ws_t* W = new pe_ws_t();
    double t_emit = 0, t_fast = 0; std::string fast; uint64_t n_fast_diff = 0;
    for (uint64_t p = 0; p < n_pairs; ++p) {
        for (int k = 0; k < 2; ++k) {
            W->off[k] = offs[2 * p + k] - offs[0]; W->m[k] = (uint32_t)(offs[2 * p + k + 1] - offs[2 * p + k]);
            W->min_score_m[k] = (int32_t)(20 + 8 * log((double)W->m[k]));
        }
        W->min_score = W->min_score_m[0] + W->min_score_m[1];
        std::vector<moni_dp_result_t> res;
        std::vector<uint32_t> cig;
        if (pe_init(*W, PP, gm.data(), rmo.data(), aux.data(), go.data(), p)) {
            pe_drive(*W, PP, nullptr, nullptr);
            while (!W->W.overflow && W->W.stage != AC_DONE) {
                std::vector<moni_dp_task_t> tasks(W->W.tasks, W->W.tasks + W->W.n_tasks);
                n_tasks += tasks.size(); ++n_rounds;
                if (be.dp(dp, tasks, res, cig)) return nullptr;
                pe_drive(*W, PP, res.data(), cig.data());
            }
        }
        if (W->W.overflow) { ++n_over; }
        if (!finalize) {
            learn[4 * p] = (!W->W.overflow && W->W.aligned) ? 1 : 0; learn[4 * p + 1] = W->final.tot; learn[4 * p + 2] = W->score2; learn[4 * p + 3] = W->final.dist;
            continue;
        }
        mh::PePairOut R;
        R.finalized = !W->W.overflow && W->W.aligned;
        R.strand = W->strand; R.tot = W->final.tot; R.score2 = W->score2; R.sub_n = W->sub_n;
        for (int k = 0; k < 2; ++k) {
            R.score2_m[k] = W->score2_m[k];
            mh::PeMateOut& M = R.mate[k];
            M.m = W->m[k]; M.off = W->off[k]; M.score = k ? W->final.m2.score : W->final.m1.score; M.filled = R.finalized && W->filled[k];
            M.ref_pos = W->ref_pos[k]; M.as = W->as[k]; M.cig = W->cigar[k]; M.n_cig = W->n_cigar[k];
            M.alt_pos = W->alt_pos[k]; M.alt_score = W->alt_score[k]; M.n_alt = W->n_alt[k]; M.orphan = W->orphan[k] != 0;
        }
        if (R.finalized && R.tot >= W->min_score) ++n_aligned;
        const size_t line_at = out.size();
        const double e0 = mh::now_s();
        mh::pe_emit(A, P, R, std::string((const char*)names + name_off[2 * p], (const char*)names + name_off[2 * p + 1]),
                    std::string((const char*)names + name_off[2 * p + 1], (const char*)names + name_off[2 * p + 2]), seq, quals, out);
        t_emit += mh::now_s() - e0;
        {   // the one-pass emitter the product runs must spell the same two lines
            const double f0 = mh::now_s();
            fast.clear();
            mh::pe_emit_fast(A, P, R, (const char*)names + name_off[2 * p], (size_t)(name_off[2 * p + 1] - name_off[2 * p]), (const char*)names + name_off[2 * p + 1],
                             (size_t)(name_off[2 * p + 2] - name_off[2 * p + 1]), seq, quals, fast);
            t_fast += mh::now_s() - f0;
            if (fast.size() != out.size() - line_at || memcmp(fast.data(), out.data() + line_at, fast.size()) != 0) {
                if (!n_fast_diff) fprintf(stderr, "host_sim pe: fast emitter differs at pair %llu:\n%s%s", (unsigned long long)p, fast.c_str(), out.c_str() + line_at);
                ++n_fast_diff;
            }
        }
    }
    delete W;
    if (getenv("MH_TIMES")) fprintf(stderr, "host_sim pe: %.3f us per pair in pe_emit, %.3f in pe_emit_fast (%llu pairs)\n", t_emit / (double)(n_pairs ? n_pairs : 1) * 1e6, t_fast / (double)(n_pairs ? n_pairs : 1) * 1e6, (unsigned long long)n_pairs);
    if (n_fast_diff) return nullptr;
    char* buf = (char*)malloc(out.size() + 1);
    memcpy(buf, out.data(), out.size() + 1);
    *out_len = out.size();
    if (stats5) { stats5[0] = n_pairs; stats5[1] = n_aligned; stats5[2] = n_tasks; stats5[3] = n_over; stats5[4] = n_rounds; }
    return buf;
}
// ---- the host pipeline for pairs (pe_big.cpp: pe_core.h with large capacities) with the CPU stand-ins for its DP batches: every pair of
// the batch goes through it ----
char* sim_align_pe_big_batch(void* s, const uint8_t* seq, const uint64_t* offs, uint64_t n_pairs, const uint8_t* names, const uint64_t* name_off,
                             const uint8_t* quals, double mean, double std_dev, int find_orphan, uint64_t* out_len, uint64_t* stats5) {
    Sim* S = (Sim*)s;
    SimBackend be;
    const uint64_t n_reads = 2 * n_pairs;
    be.S = S; be.seq = seq; be.offs = offs; be.n_reads = n_reads;
    moni_align_params_t P;
    memset(&P, 0, sizeof P);
    P.min_len = 25; P.ext_len = 100; P.check_k = 5; P.region_dist = 10; P.filter_seeds = 1; P.n_seeds_thr = 1000; P.filter_freq = 1;
    P.left_mem_check = 1; P.freq_thr = 0.5; P.smatch = 2; P.smismatch = 4; P.gapo = 4; P.gapo2 = 13; P.gape = 2; P.gape2 = 1;
    P.end_bonus = 400; P.w = -1; P.zdrop = -1; P.max_dist_x = 500; P.max_dist_y = 100; P.max_iter = 10; P.max_pred = 5;
    P.min_chain_score = 40; P.min_chain_length = 1; P.host_threads = 1;
    moni_seed_params_t sp{P.min_len, P.filter_seeds, P.n_seeds_thr, 0};
    std::vector<moni_mem_t> gm; std::vector<uint64_t> go, rmo;
    if (be.seed(sp, gm, go, rmo)) return nullptr;
    pe_params_t PP;
    memset(&PP, 0, sizeof PP);
    ac_params_t& AP = PP.P;
    AP.min_len = P.min_len; AP.ext_len = P.ext_len; AP.check_k = P.check_k; AP.region_dist = P.region_dist; AP.filter_freq = P.filter_freq;
    AP.left_mem_check = P.left_mem_check; AP.freq_thr = P.freq_thr; AP.smatch = P.smatch; AP.gapo = P.gapo; AP.gapo2 = P.gapo2; AP.gape = P.gape;
    AP.gape2 = P.gape2; AP.max_dist_x = P.max_dist_x; AP.max_dist_y = P.max_dist_y; AP.max_iter = P.max_iter; AP.max_pred = P.max_pred;
    AP.min_chain_score = P.min_chain_score; AP.min_chain_length = P.min_chain_length; AP.n_text = S->hix.n_text; AP.n_seq = (uint32_t)S->hix.names.size();
    AP.seq_starts = S->hix.seq_starts.data();
    AP.lift_seqs = S->hix.lift_seqs.data(); AP.lift_runs = S->hix.lift_runs.data(); AP.pdir = nullptr;          // as moni_pe_align_batch hands it over
    PP.smismatch = P.smismatch; PP.max_penalty = std::max(P.smatch + P.smismatch, P.gapo + P.gape); PP.filter_dir = 1; PP.finalize = 1;
    PP.dir_thr = 50.0; PP.mean = (float)mean; PP.std_dev = (float)std_dev;
    PP.find_orphan = (find_orphan & 1) ? 1 : 0; PP.secondary_chains = (find_orphan & 2) ? 1 : 0; PP.w = (uint32_t)S->hix.w; PP.ins_mean = mean; PP.ins_std_dev = std_dev;      // bit 1: -Z
    moni_dp_params_t dp;
    memset(&dp, 0, sizeof dp);
    dp.m = 5;
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) dp.mat[i * 5 + j] = i == j ? P.smatch : (int8_t)-P.smismatch; }
    dp.q = P.gapo; dp.e = P.gape; dp.w = -1; dp.zdrop = -1; dp.end_bonus = P.end_bonus;
    std::vector<uint64_t> rel(n_reads + 1);
    for (uint64_t r = 0; r <= n_reads; ++r) rel[r] = offs[r] - offs[0];
    std::vector<PeBigPair> pairs(n_pairs);
    for (uint64_t p = 0; p < n_pairs; ++p) pairs[p].pair = p;
    uint64_t n_tasks = 0, n_rounds = 0;
    const int rc = pe_big_run(&PP, sizeof PP, gm.data(), rmo.data(), S->aux.data(), go.data(), rel.data(), pairs,
                              [&](const std::vector<moni_dp_task_t>& t, std::vector<moni_dp_result_t>& r, std::vector<uint32_t>& cg) -> int { n_tasks += t.size(); ++n_rounds; return be.dp(dp, t, r, cg); });
    if (rc) return nullptr;
    mh::Aligner A(S->hix, P, seq, offs);
    std::string out;
    uint64_t n_aligned = 0, n_over = 0;
    for (uint64_t p = 0; p < n_pairs; ++p) {
        const PeBigPair& B = pairs[p];
        if (B.status == 2) ++n_over;
        mh::PePairOut R;
        R.finalized = B.status == 1; R.strand = B.strand; R.tot = B.tot; R.score2 = B.score2; R.sub_n = B.sub_n;
        int32_t ms = 0;
        for (int k = 0; k < 2; ++k) {
            R.score2_m[k] = B.score2_m[k];
            mh::PeMateOut& M = R.mate[k];
            M.m = (uint32_t)(rel[2 * p + k + 1] - rel[2 * p + k]); M.off = rel[2 * p + k]; M.score = B.mate_score[k]; M.filled = R.finalized && B.filled[k];
            M.ref_pos = B.ref_pos[k]; M.as = B.as[k]; M.cig = B.cig[k].data(); M.n_cig = (uint32_t)B.cig[k].size();
            M.alt_pos = B.alt_pos[k].data(); M.alt_score = B.alt_score[k].data(); M.n_alt = (uint32_t)B.alt_pos[k].size(); M.orphan = B.orphan[k] != 0;
            ms += (int32_t)(20 + 8 * log((double)M.m));
        }
        if (R.finalized && R.tot >= ms) ++n_aligned;
        mh::pe_emit_fast(A, P, R, (const char*)names + name_off[2 * p], (size_t)(name_off[2 * p + 1] - name_off[2 * p]), (const char*)names + name_off[2 * p + 1],
                         (size_t)(name_off[2 * p + 2] - name_off[2 * p + 1]), seq, quals, out);
    }
    char* buf = (char*)malloc(out.size() + 1);
    memcpy(buf, out.data(), out.size() + 1);
    *out_len = out.size();
    if (stats5) { stats5[0] = n_pairs; stats5[1] = n_aligned; stats5[2] = n_tasks; stats5[3] = n_over; stats5[4] = n_rounds; }
    return buf;
}
void sim_free(void* p) { free(p); }

// ---- the product's lift tables (lift_build.hpp + lift_core.h) on their own: positions and CIGARs ----
struct LiftSim { LiftTables lt; std::vector<uint64_t> seq_starts; };
void* liftsim_create(const moni_flat_index_t* f) {
    LiftSim* L = new LiftSim();
    std::string err;
    if (L->lt.build(*f, err)) { fprintf(stderr, "host_sim: %s\n", err.c_str()); delete L; return nullptr; }
    L->seq_starts.assign(f->seq_starts, f->seq_starts + f->n_seq + 1);
    return L;
}
void liftsim_destroy(void* h) { delete (LiftSim*)h; }
uint64_t liftsim_n_runs(void* h, uint64_t seq) { return ((LiftSim*)h)->lt.seqs[seq].n_runs; }
static const moni_lift_seq_t& liftsim_seq(LiftSim* L, uint64_t pos, uint64_t& start) {
    const size_t rk = (size_t)(std::lower_bound(L->seq_starts.begin(), L->seq_starts.end(), pos + 1) - L->seq_starts.begin());
    start = pos - L->seq_starts[rk - 1];
    return L->lt.seqs[rk - 1];
}
void liftsim_lift(void* h, const uint64_t* pos, uint64_t n, uint64_t* out) {
    LiftSim* L = (LiftSim*)h;
    for (uint64_t i = 0; i < n; ++i) { uint64_t st; const moni_lift_seq_t& S = liftsim_seq(L, pos[i], st); out[i] = S.second + lift_pos(L->lt.runs.data() + S.run_off, S.n_runs, st); }
}
int64_t liftsim_cigar(void* h, uint64_t pos, const uint32_t* cig, uint32_t n_cig, uint32_t* out, uint32_t cap) {
    LiftSim* L = (LiftSim*)h;
    uint64_t st; const moni_lift_seq_t& S = liftsim_seq(L, pos, st);
    return lift_cigar(L->lt.runs.data() + S.run_off, S.n_runs, st, cig, n_cig, out, cap);
}

}  // extern "C"
