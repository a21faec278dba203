// libmoni_hip.so: C ABI (include/moni_hip.h) over the HIP kernels.  Host side only orchestrates:
// device buffers, launches on the context's stream, exclusive scans (rocPRIM), HIP-event timing.
// There is no CPU implementation of any kernel in this library: without a HIP device every entry
// point fails with MONI_ENODEV.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstring>
#include <atomic>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>

#include <sched.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <algorithm>
#include <vector>

#include "../../include/moni_hip.h"
#include "image.hpp"
#include "lift_build.hpp"
#include "ref_index_io.hpp"
#include "ms_index_io.hpp"
#include "align_host.hpp"
#include "layout.h"
#include "seed_kernels.hip"
#include "extz_kernels.hip"
#include "align_kernel.hip"
#include "pe_kernel.hip"
#include "pe_host.hpp"
#include "pe_big.h"
#include "align_fast.hip"
#include "pe_fast.hip"
#include "pe_lines.hip"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "moni_hip: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); return MONI_ENODEV; } } while (0)

enum { EV_MS0 = 0, EV_MS1, EV_MC0, EV_MC1, EV_ME0, EV_ME1, EV_PC0, EV_PC1, EV_PE0, EV_PE1, EV_DP0, EV_DP1, EV_ALL0, EV_ALL1, EV_N };

struct moni_index {
    int device = 0;
    moni_consts_t K;
    moni_tables_t* d_tables = nullptr;
    moni_row_t* d_rows = nullptr;
    moni_frow_t* d_frows = nullptr;
    uint32_t* d_cr = nullptr;
    moni_rec_t* d_recs = nullptr;
    moni_phi_t *d_phi = nullptr, *d_phi_inv = nullptr;
    uint32_t *d_phi_dir = nullptr, *d_phi_inv_dir = nullptr;
    uint8_t* d_text = nullptr;
    uint64_t* d_text2 = nullptr; uint32_t* d_exc = nullptr; uint32_t exc_sh = 10, exc_words = 0;      // 2-bit text and its exception bitmap (seed_core.h: mem_fast_t)
    uint64_t* d_seq_starts = nullptr;
    uint32_t* d_name_id = nullptr;
    uint8_t* d_snames = nullptr; uint32_t* d_sname_off = nullptr;      // sequence names, ragged (SAM text in align_kernel)
    moni_lift_seq_t* d_lift_seqs = nullptr; moni_lift_run_t* d_lift_runs = nullptr;   // liftidx::lifts (lift_core.h)
    uint64_t* d_pdir = nullptr;
    bool lifts_null = true;
    uint64_t bytes = 0;
    // host copies for the host stages of the full path (chaining, MD/NM, SAM)
    std::vector<uint8_t> h_text;
    mh::HostIndex hix;
};

template <class Tp>
struct DBuf {
    Tp* p = nullptr;
    size_t cap = 0;
    int ensure(size_t need) {
        if (need <= cap) return MONI_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = need + need / 8 + 64;
        if (hipMalloc((void**)&p, want * sizeof(Tp)) != hipSuccess) { p = nullptr; return MONI_ENOMEM; }
        cap = want;
        return MONI_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

template <class Tp>
struct HBuf {                  // pinned host buffer, grow-only
    Tp* p = nullptr; size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return MONI_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = n + n / 4;
        if (hipHostMalloc((void**)&p, want * sizeof(Tp), hipHostMallocDefault) != hipSuccess) { p = nullptr; return MONI_ENOMEM; }
        cap = want; return MONI_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

#define PE_NSET 1          // sets of the staged paired kernels' large buffers (pe_api.inc): the chunks' staged kernels run one after the other on one stream, what outlives them is per chunk
#define PE_NSTREAM 4       // hand-over launches of the paired path in flight, one stream and one slot array each
#ifndef AK_NSET
#define AK_NSET 2
#endif                     // launch streams of the align stage, each with its own buffer set: sub-batch k runs on set k % AK_NSET
struct moni_ctx {
    moni_index* idx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[EV_N];
    bool ev_valid[EV_N];
    // resident batch
    DBuf<uint8_t> seq;
    DBuf<uint64_t> offs;
    uint64_t n_reads = 0, total_len = 0, max_len = 0;
    DBuf<moni_u64x2> blk; std::vector<moni_u64x2> h_blk;      // workspace layout of the resident batch (seed_core.h: ws_ptr_base / ws_pat_base), n_blocks + 1 entries
    std::vector<uint8_t> h_seq;              // host copy of the resident batch (SAM SEQ, MD/NM)
    std::vector<uint64_t> h_offs;
    struct Stash { DBuf<uint8_t> seq; DBuf<uint64_t> offs; DBuf<moni_u64x2> blk; std::vector<moni_u64x2> h_blk; uint64_t n_reads = 0, total_len = 0, max_len = 0; std::vector<uint8_t> h_seq; std::vector<uint64_t> h_offs; };
    std::vector<Stash> stash;                // moni_reads_swap: further batches kept in HBM beside the resident one
    // workspaces
    DBuf<uint64_t> ptr, pat; DBuf<uint8_t> pflag;      // pflag: per seeding task, its pattern holds a byte outside A / C / G / T (pack_kernel)
    DBuf<uint32_t> cnt_m, cnt_s;
    DBuf<moni_u64x2> mem_slots;
    DBuf<uint64_t> tot, read_mem_off;
    DBuf<moni_mem_t> mems;
    DBuf<uint32_t> aux;
    DBuf<uint64_t> lowers, tmp, occ_cnt, occ_off, occs;
    DBuf<uint32_t> pool;
    DBuf<uint8_t> scan_tmp;
    uint32_t* d_small = nullptr;            // [0] pool_next, [1] error_flag
    unsigned long long* d_counters = nullptr;   // 4
    uint64_t n_mems = 0, n_occs = 0;
    uint32_t dirs_scale = 1;                                   // direction-bit budget multiplier (doubles after a batch that overflowed it)
    uint32_t tmp_cap = 16;
    uint32_t pool_rows = 4096;
    double dp_kernel_ms_accum = 0;
    int extz_lds = 0;                          // moni_extz_batch through the LDS-tiled kernel (MONI_EXTZ_LDS=1)
    int ms_variant = 0;                        // 0 = default; MONI_MS_VARIANT selects an (interleave, occupancy) variant for tuning
    // dp
    DBuf<uint8_t> dp_q, dp_t, dp_dir;
    DBuf<moni_dp_task_t> dp_tasks;
    DBuf<moni_dp_result_t> dp_res;
    DBuf<uint32_t> dp_cig;
    DBuf<uint64_t> dp_off;
    DBuf<uint32_t> dp_ws;
    // align kernel
    DBuf<dp_big_t> dp_big;
    DBuf<uint8_t> dp_dir_big;
    struct AfSet {          // device buffers of the staged align kernels (align_fast.hip), one set per launch stream
        DBuf<af_plan_t> plans; DBuf<moni_dp_task_t> tasks; DBuf<af_res_t> res; DBuf<uint32_t> bin_q, task_pos, tb_task, big_list, ctr; DBuf<uint8_t> ntasks;
        DBuf<af_chunk_t> chunks; DBuf<uint8_t> dirs, fin; DBuf<uint32_t> recipes; DBuf<af_ctab_t> ctab; DBuf<af_tb_t> tb; DBuf<uint64_t> bnd; DBuf<unsigned long long> prof, txt_cur;
        void release() { ctab.release(); ntasks.release(); bnd.release(); prof.release(); big_list.release(); txt_cur.release(); plans.release(); tasks.release(); res.release(); bin_q.release(); task_pos.release(); tb_task.release(); ctr.release();
                         chunks.release(); dirs.release(); fin.release(); recipes.release(); tb.release(); }
    } af[AK_NSET], af_pe[PE_NSET];
    HBuf<unsigned long long> pe_hcur;       // paired path, per chunk: the pool cursors / DP counters (8 words) and the number of pairs handed over, copied behind the chunk's kernels
    HBuf<uint32_t> af_ctr_host;             // counters of the last batch's launches (64 words per sub-batch), pinned
    DBuf<uint32_t> fb_all;                  // per sub-batch: the number of reads handed to align_kernel (16 words apart), then their lists
    DBuf<ak_slot_t> ak_slots;
    DBuf<ak_wave_t> ak_waves;
    DBuf<unsigned long long> ak_cursors;
    bool mapq_tab_ready = false;
    DBuf<uint8_t> ak_rnames, ak_quals; DBuf<uint64_t> ak_rname_off, ak_txt; DBuf<double> ak_mapq_tab;      // SAM text in the kernel
    HBuf<uint64_t> h_txt;                                 // pinned staging of one sub-batch's text
    uint64_t ak_waves_full = 0;
    hipStream_t ak_stream[AK_NSET] = {}, copy_stream = nullptr, fb_stream[AK_NSET] = {}, pe_stream[PE_NSTREAM] = {};
    std::vector<hipEvent_t> ak_fin;                // staged kernels of a sub-batch queued; the handed-over reads follow on fb_stream
    std::vector<hipEvent_t> af_ev;                 // per sub-batch: after the chaining kernels, after the DP kernels, after selection + traceback (HIP-event kernel times)
    std::vector<hipEvent_t> ak_begin, ak_done;
    HBuf<moni_aln_rec_t> h_recs; HBuf<uint32_t> h_cig; HBuf<moni_alt_t> h_alt; HBuf<uint64_t> h_md;      // pinned staging of one sub-batch's records
    std::vector<mh::Aligner::OutBuf> pieces;          // per host thread: the text it is writing (kept across batches)
    std::vector<std::vector<char>> md_scratch;
    DBuf<moni_aln_rec_t> ak_recs;
    DBuf<uint32_t> ak_cig;
    DBuf<moni_alt_t> ak_alt;
    DBuf<int32_t> ak_minscore;
    struct PeBufs { DBuf<pe_slot_t> slots; DBuf<ak_wave_t> waves; DBuf<pe_rec_t> recs; DBuf<uint32_t> cig; DBuf<moni_alt_t> alt; DBuf<unsigned long long> cur; DBuf<int32_t> minscore;
                    DBuf<pe_sel_t> sel[PE_NSET]; DBuf<uint32_t> fb[PE_NSET];
                    DBuf<uint64_t> txt_pool, block, dev_len, dev_off, dev_pos; DBuf<int32_t> subn_tab; DBuf<double> pen_tab; DBuf<pe_pslot_t> park, k1_slots; DBuf<uint8_t> k1_dirs; DBuf<uint32_t> parked; DBuf<pe_orec_t> orec; uint32_t tag = 0;      // pe_orphan_kernel: parked pairs, their chains' scores
                    DBuf<uint8_t> scan_tmp[PE_NSTREAM]; bool subn_ready = false;      // the lines written on the GPU (pe_lines.hip)
                             // staged paired kernels (pe_fast.hip): per chunk in flight, what pe_select_kernel decided; the hand-over list (16 words of counter, then the pairs)
                    void release() { slots.release(); waves.release(); recs.release(); cig.release(); alt.release(); cur.release(); minscore.release(); for (int x = 0; x < PE_NSET; ++x) { sel[x].release(); fb[x].release(); }
                                     txt_pool.release(); block.release(); dev_len.release(); dev_off.release(); dev_pos.release(); subn_tab.release(); pen_tab.release(); park.release(); k1_slots.release(); k1_dirs.release(); parked.release(); orec.release(); for (int x = 0; x < PE_NSTREAM; ++x) scan_tmp[x].release(); } } pe;      // paired-end path (pe_api.inc)
    unsigned long long* d_ak_cursors = nullptr;
    char* out_buf = nullptr; size_t out_cap = 0;      // moni_align_run's text buffer, kept across calls; pinned (hipHostMalloc): the in-order blocks of the
                                                      // sub-batches land in it by DMA
    DBuf<uint64_t> ak_block;                          // per sub-batch: its SAM lines in read order (gather_lines_kernel)
    DBuf<uint64_t> ak_dev_len, ak_dev_off, ak_dev_pos; DBuf<unsigned long long> ak_dev_sum;
    HBuf<unsigned long long> h_sum;                   // per sub-batch: bytes of the block, records that need the host, aligned reads
    DBuf<uint8_t> gather_tmp[AK_NSET];                      // rocPRIM scan workspace of the gather, one per stream it runs on (the two streams' scans overlap)
    float ak_kernel_ms = 0;
    int n_cu_cached = 0, pe_occ_cached = 0;          // hipGetDeviceProperties / the occupancy query take a millisecond each: asked once per context
};

namespace {

template <class Tp>
int upload(Tp** d, const std::vector<Tp>& h, uint64_t& bytes) {
    size_t nb = h.size() * sizeof(Tp);
    if (hipMalloc((void**)d, nb ? nb : 8) != hipSuccess) return MONI_ENOMEM;
    if (nb && hipMemcpy(*d, h.data(), nb, hipMemcpyHostToDevice) != hipSuccess) return MONI_ENODEV;
    bytes += nb;
    return MONI_OK;
}

int ctx_n_cu(moni_ctx* c) {
    if (!c->n_cu_cached) { hipDeviceProp_t pr; c->n_cu_cached = hipGetDeviceProperties(&pr, c->idx->device) == hipSuccess ? pr.multiProcessorCount : 256; }
    return c->n_cu_cached;
}
int exclusive_scan_u64(moni_ctx* c, uint64_t* in, uint64_t* out, size_t n) {
    size_t tmp_bytes = 0;
    if (rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c->stream) != hipSuccess) return MONI_ENODEV;
    int rc = c->scan_tmp.ensure(tmp_bytes + 16);
    if (rc) return rc;
    if (rocprim::exclusive_scan(c->scan_tmp.p, tmp_bytes, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c->stream) != hipSuccess) return MONI_ENODEV;
    return MONI_OK;
}

inline void rec(moni_ctx* c, int e) { (void)hipEventRecord(c->ev[e], c->stream); c->ev_valid[e] = true; }

}  // namespace

// ---- the text from the BWT -------------------------------------------------------------------------------------------------------------
// The aligner's constructor takes the text from <prefix>.plain.slp (seed_finder.hpp:88-99), a ShapedSlp grammar whose format is not
// available here; but the text is redundant with the r-index: BWT[p] = T[SA[p] - 1] and LF(p) is the position of SA[p] - 1, so a walk
// that starts at a sampled position p (a run boundary: SA[p] - 1 is stored) spells the text backwards from there.  The 2 r samples
// (samples_start, samples_last), sorted, cut the text into 2 r pieces; one lane per piece walks LF with the general rows of the move
// structure from its sample down to the next smaller one and writes the run heads it passes.  keys / vals: the samples in increasing
// order and where they sit in the BWT (run << 1 | 1 for the last position of the run).
__global__ void __launch_bounds__(256) text_rebuild_kernel(const moni_row_t* __restrict__ rows, uint64_t r, const uint8_t* __restrict__ heads, const uint64_t* __restrict__ keys,
                                                           const uint64_t* __restrict__ vals, uint64_t n_samples, uint8_t* __restrict__ text, uint64_t n_text,
                                                           unsigned long long* __restrict__ n_written) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long wrote = 0;
    if (i < n_samples) {
        // a sample is the text position OF the BWT symbol at its place (SA - 1 mod n: moni.hpp:127-128), so T[sample] = that symbol; the
        // piece of this lane: positions s, s - 1, ... down to one above the next smaller sample (the smallest sample: down to 0)
        const uint64_t s = keys[i];
        const uint64_t count = i ? s - keys[i - 1] : s + 1;
        if (count && s <= n_text) {
            const uint64_t v = vals[i];
            uint32_t run = (uint32_t)(v >> 1);
            moni_row_t A = ld_row(rows, run);
            uint64_t pos = (v & 1) ? ld_start(rows, run + 1) - 1 : row_start(A);
            for (uint64_t t = 0; t < count; ++t) {               // T[s - t] = BWT[pos], then pos = LF(pos)
                const uint64_t j = s - t;
                if (j < n_text) { text[j] = heads[run]; ++wrote; }          // (position n - 1 is the terminator's: not part of the text)
                if (t + 1 == count) break;
                pos = row_lfbase(A) + (pos - row_start(A));
                run = row_dest(A);
                settle_run(rows, r, pos, run, A);
            }
        }
    }
    wave_add(wrote, n_written);
}

// f.text == NULL: the text rebuilt on the device from the rows already uploaded (I->d_rows); checked against the BWT's own symbol counts
static int rebuild_text(moni_index* I, const moni_flat_index_t& f, std::vector<uint8_t>& text) {
    const uint64_t r = f.r, n_text = f.n - 1, ns = 2 * r;
    uint64_t *d_k[2] = {nullptr, nullptr}, *d_v[2] = {nullptr, nullptr}; uint8_t *d_heads = nullptr, *d_text = nullptr; void* d_tmp = nullptr; unsigned long long* d_cnt = nullptr;
    auto done = [&](int code) { void* ps[] = {d_k[0], d_k[1], d_v[0], d_v[1], d_heads, d_text, d_tmp, d_cnt}; for (void* p : ps) if (p) (void)hipFree(p); return code; };
    try {
        std::vector<uint64_t> hk(ns), hv(ns);
        for (uint64_t k = 0; k < r; ++k) { hk[2 * k] = f.ssa[k]; hv[2 * k] = k << 1; hk[2 * k + 1] = f.esa[k]; hv[2 * k + 1] = (k << 1) | 1; }
        for (int x = 0; x < 2; ++x) if (hipMalloc((void**)&d_k[x], ns * 8) != hipSuccess || hipMalloc((void**)&d_v[x], ns * 8) != hipSuccess) return done(MONI_ENOMEM);
        if (hipMalloc((void**)&d_heads, r + 8) != hipSuccess || hipMalloc((void**)&d_text, n_text + 16) != hipSuccess || hipMalloc((void**)&d_cnt, 8) != hipSuccess) return done(MONI_ENOMEM);
        if (hipMemcpy(d_k[0], hk.data(), ns * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_v[0], hv.data(), ns * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_heads, f.heads, r, hipMemcpyHostToDevice) != hipSuccess || hipMemset(d_cnt, 0, 8) != hipSuccess || hipMemset(d_text, 0, n_text + 16) != hipSuccess) return done(MONI_ENODEV);
        size_t tmp_bytes = 0;
        if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_k[0], d_k[1], d_v[0], d_v[1], (size_t)ns, 0u, 40u, (hipStream_t)0) != hipSuccess) return done(MONI_ENODEV);
        if (hipMalloc(&d_tmp, tmp_bytes + 16) != hipSuccess) return done(MONI_ENOMEM);
        if (rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_k[0], d_k[1], d_v[0], d_v[1], (size_t)ns, 0u, 40u, (hipStream_t)0) != hipSuccess) return done(MONI_ENODEV);
        hipLaunchKernelGGL(text_rebuild_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, 0, I->d_rows, r, d_heads, d_k[1], d_v[1], ns, d_text, n_text, d_cnt);
        unsigned long long wrote = 0;
        text.resize(n_text);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&wrote, d_cnt, 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(text.data(), d_text, n_text, hipMemcpyDeviceToHost) != hipSuccess) return done(MONI_ENODEV);
        if (wrote != n_text) { fprintf(stderr, "moni_hip: the BWT walk wrote %llu of %llu text positions: samples and runs disagree\n", wrote, (unsigned long long)n_text); return done(MONI_ERANGE); }
        // every symbol as often as the BWT holds it (the terminator is stored as byte 1 in the run heads and is not part of the text)
        uint64_t cnt[256] = {0}, bw[256] = {0};
        for (uint64_t i = 0; i < n_text; ++i) cnt[text[i]]++;
        for (uint64_t k = 0; k < r; ++k) bw[f.heads[k]] += f.starts[k + 1] - f.starts[k];
        cnt[1]++;
        for (int b = 0; b < 256; ++b) if (cnt[b] != bw[b]) { fprintf(stderr, "moni_hip: the rebuilt text holds byte %d %llu times, the BWT %llu times\n", b, (unsigned long long)cnt[b], (unsigned long long)bw[b]); return done(MONI_ERANGE); }
        return done(MONI_OK);
    } catch (const std::bad_alloc&) { return done(MONI_ENOMEM); }
}

// The paired path keeps ~11 streams busy (staged kernels, transfers, up to PE_NSTREAM hand-over kernels of ~30 ms each); streams beyond the runtime's
// hardware queues share one, and a kernel behind a hand-over kernel on the same queue waits for it (8 queues: 214 ms per 1 M pairs, 16: 194 ms,
// profiles/r03o).  The runtime reads GPU_MAX_HW_QUEUES when it initialises: set a default when the process has none (never overrides the caller's).
__attribute__((constructor)) static void moni_hip_default_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

extern "C" {

const char* moni_version(void) { return "moni_hip 0.1 (gfx950)"; }

int moni_index_create(const moni_flat_index_t* f, int device, moni_index_t** out) {
    if (!f || !out || !f->F || !f->heads || !f->starts || !f->ssa || !f->esa || !f->thr || !f->seq_starts)
        return MONI_EINVAL;              // f->text may be NULL: the text is then rebuilt from the BWT on the device (rebuild_text);
                                         // f->slcp may be NULL: no LCP samples (`-n`), the occurrence walks measure the LCP on the text
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) {
        fprintf(stderr, "moni_hip: no HIP device %d (found %d); this library has no CPU path\n", device, ndev);
        return MONI_ENODEV;
    }
    HIPCHK(hipSetDevice(device));
    if (f->n < 2 || f->r < 1 || f->n_seq < 1) return MONI_EINVAL;
    HostImage img;
    int rc = img.build(*f);
    if (rc) { fprintf(stderr, "moni_hip: index rejected: %s\n", img.err.c_str()); return rc; }
    LiftTables lt;
    { std::string lerr; if ((rc = lt.build(*f, lerr))) { fprintf(stderr, "moni_hip: index rejected: %s\n", lerr.c_str()); return rc; } }
    moni_index* I = new moni_index();
    I->device = device;
    I->K = img.K;
    I->lifts_null = lt.all_null;
    std::vector<moni_tables_t> tv(1, img.T);
    std::vector<uint8_t> text;
    if (f->text) text.assign(f->text, f->text + (f->n - 1));
    else {
        if ((rc = upload(&I->d_rows, img.rows, I->bytes)) || (rc = rebuild_text(I, *f, text))) { moni_index_destroy(I); return rc; }
    }
    I->h_text = text;
    text.resize(text.size() + 16, 0);            // text_byte() reads aligned 8-byte words
    std::vector<uint32_t> name_id(f->n_seq);
    {   // names; sequences that share a name share a counter in the per-genome filter (std::map<std::string,...>, seed_finder.hpp:331-343)
        const char* p = f->seq_names;
        for (uint64_t i = 0; i < f->n_seq; ++i) {
            std::string nm = p ? std::string(p) : ("seq" + std::to_string(i));
            if (p) p += nm.size() + 1;
            name_id[i] = (uint32_t)i;
            for (uint64_t j = 0; j < i; ++j) if (I->hix.names[j] == nm) { name_id[i] = name_id[j]; break; }
            I->hix.names.push_back(nm);
        }
    }
    std::vector<uint8_t> sname_blob; std::vector<uint32_t> sname_off(1, 0);
    for (const auto& nm : I->hix.names) { sname_blob.insert(sname_blob.end(), nm.begin(), nm.end()); sname_off.push_back((uint32_t)sname_blob.size()); }
    sname_blob.resize(sname_blob.size() + 8, 0);
    I->hix.n_text = f->n - 1; I->hix.w = f->w; I->hix.text = I->h_text.data();
    I->hix.seq_starts.assign(f->seq_starts, f->seq_starts + f->n_seq + 1);
    I->hix.lift_seqs = lt.seqs; I->hix.lift_runs = lt.runs;
    if ((rc = upload(&I->d_pdir, lt.pdir, I->bytes)) || (rc = upload(&I->d_lift_seqs, lt.seqs, I->bytes)) || (rc = upload(&I->d_lift_runs, lt.runs, I->bytes)) || (rc = upload(&I->d_tables, tv, I->bytes)) || (!I->d_rows && (rc = upload(&I->d_rows, img.rows, I->bytes))) || (rc = upload(&I->d_frows, img.frows, I->bytes)) ||
        (rc = upload(&I->d_cr, img.cr, I->bytes)) || (rc = upload(&I->d_recs, img.recs, I->bytes)) ||
        (rc = upload(&I->d_phi, img.phi, I->bytes)) || (rc = upload(&I->d_phi_inv, img.phi_inv, I->bytes)) ||
        (rc = upload(&I->d_phi_dir, img.phi_dir, I->bytes)) || (rc = upload(&I->d_phi_inv_dir, img.phi_inv_dir, I->bytes)) ||
        (rc = upload(&I->d_text, text, I->bytes)) || (rc = upload(&I->d_seq_starts, img.seq_starts, I->bytes)) ||
        (rc = upload(&I->d_name_id, name_id, I->bytes)) || (rc = upload(&I->d_snames, sname_blob, I->bytes)) || (rc = upload(&I->d_sname_off, sname_off, I->bytes))) {
        moni_index_destroy(I);
        return rc;
    }
    {   // the 2-bit text of mem_kernel's comparison loop, from the byte text on the device
        const uint64_t n_text = f->n - 1, n_words = n_text / 32 + 2;
        I->exc_sh = text2_exc_shift(n_text, MONI_EXC_BITS);
        I->exc_words = (uint32_t)((((n_text >> I->exc_sh) + 1) + 31) / 32);
        if (hipMalloc((void**)&I->d_text2, n_words * 8) != hipSuccess || hipMalloc((void**)&I->d_exc, (size_t)I->exc_words * 4 + 4) != hipSuccess) { moni_index_destroy(I); return MONI_ENOMEM; }
        I->bytes += n_words * 8 + (uint64_t)I->exc_words * 4;
        if (hipMemset(I->d_exc, 0, (size_t)I->exc_words * 4 + 4) != hipSuccess) { moni_index_destroy(I); return MONI_ENODEV; }
        hipLaunchKernelGGL(text2_build_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, 0, I->d_text, n_text, n_words, I->d_text2, I->d_exc, I->exc_sh);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { moni_index_destroy(I); return MONI_ENODEV; }
    }
    *out = I;
    return MONI_OK;
}

int moni_index_load(const char* path, int device, moni_index_t** out) {
    if (!path || !out) return MONI_EINVAL;
    // The file is mapped read-only and the arrays are used where they lie (every section starts on an 8-byte boundary): no private copy of the ~10 GB,
    // and processes that load the same file - one rank per GPU - share its pages.
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return MONI_EIO;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 56) { ::close(fd); return MONI_EIO; }
    const uint64_t fsize = (uint64_t)sb.st_size;
    void* map = mmap(nullptr, fsize, PROT_READ, MAP_SHARED, fd, 0);
    ::close(fd);
    if (map == MAP_FAILED) return MONI_EIO;
    struct Unmap { void* p; uint64_t n; ~Unmap() { munmap(p, n); } } unmap{map, fsize};
    const uint8_t* base = static_cast<const uint8_t*>(map);
    if (memcmp(base, "MONIFLT2", 8) != 0) return MONI_EIO;
    uint64_t hdr[6];
    memcpy(hdr, base + 8, 48);
    const uint64_t n = hdr[0], r = hdr[1], w = hdr[2], nseq = hdr[3], nblob = hdr[4];
    const bool has_lifts = hdr[5] != 0;
    // the header is not trusted: every count is checked against the file size before anything is used
    auto pad8 = [](uint64_t b) { return (b + 7) & ~7ull; };
    if (n < 2 || r < 1 || r > n || nseq < 1 || nseq > n || n > fsize || nblob > fsize ||
        56 + 256 * 8 + pad8(r) + (r + 1) * 8 + 4 * r * 8 + pad8(n - 1) + (nseq + 1) * 8 + pad8(nblob) > fsize) return MONI_EIO;
    try {
        uint64_t at = 56;
        bool ok = true;
        auto take = [&](uint64_t bytes) -> const uint8_t* {
            if (!ok || bytes > fsize || at + pad8(bytes) > fsize + 7 || at + bytes > fsize) { ok = false; return base; }
            const uint8_t* p = base + at;
            at += pad8(bytes);
            return p;
        };
        const uint64_t* F = reinterpret_cast<const uint64_t*>(take(256 * 8));
        const uint8_t* heads = take(r);
        const uint64_t* starts = reinterpret_cast<const uint64_t*>(take((r + 1) * 8));
        const uint64_t* ssa = reinterpret_cast<const uint64_t*>(take(r * 8));
        const uint64_t* esa = reinterpret_cast<const uint64_t*>(take(r * 8));
        const uint64_t* thr = reinterpret_cast<const uint64_t*>(take(r * 8));
        const uint64_t* slcp = reinterpret_cast<const uint64_t*>(take(r * 8));
        const uint8_t* text = take(n - 1);
        const uint64_t* seq_starts = reinterpret_cast<const uint64_t*>(take((nseq + 1) * 8));
        const char* blob = reinterpret_cast<const char*>(take(nblob));
        if (!ok) return MONI_EIO;
        // lifts (liftidx.hpp:131-143): per sequence second, columns, #ins, #del, then the two lists of ones
        std::vector<uint64_t> l_second, l_len, l_ins_off(1, 0), l_del_off(1, 0), l_ins, l_del;
        if (has_lifts) {
            for (uint64_t i = 0; i < nseq && ok; ++i) {
                uint64_t h4[4];
                const uint8_t* hp = take(32);
                if (!ok) break;
                memcpy(h4, hp, 32);
                if (h4[2] > fsize / 8 || h4[3] > fsize / 8) { ok = false; break; }
                l_second.push_back(h4[0]); l_len.push_back(h4[1]);
                const uint64_t* a = reinterpret_cast<const uint64_t*>(take(h4[2] * 8));
                const uint64_t* d = reinterpret_cast<const uint64_t*>(take(h4[3] * 8));
                if (!ok) break;
                l_ins.insert(l_ins.end(), a, a + h4[2]); l_del.insert(l_del.end(), d, d + h4[3]);
                l_ins_off.push_back(l_ins.size()); l_del_off.push_back(l_del.size());
            }
        }
        if (!ok) return MONI_EIO;
        std::vector<char> names_flat;
        for (uint64_t i = 0, p = 0; i < nseq; ++i) {
            uint64_t ln;
            if (p + 8 > nblob) return MONI_EIO;
            memcpy(&ln, blob + p, 8);
            if (ln > nblob || p + 8 + ln > nblob) return MONI_EIO;
            names_flat.insert(names_flat.end(), blob + p + 8, blob + p + 8 + ln);
            names_flat.push_back(0);
            p += 8 + ln;
        }
        moni_flat_index_t f;
        memset(&f, 0, sizeof f);
        f.n = n; f.r = r; f.w = w; f.n_seq = nseq;
        f.F = F; f.heads = heads; f.starts = starts; f.ssa = ssa; f.esa = esa;
        f.thr = thr; f.slcp = slcp; f.text = text; f.seq_starts = seq_starts; f.seq_names = names_flat.data();
        l_ins.push_back(0); l_del.push_back(0);      // never empty: the pointers stay valid
        if (has_lifts) {
            f.lift_second = l_second.data(); f.lift_len = l_len.data(); f.lift_ins_off = l_ins_off.data(); f.lift_ins = l_ins.data();
            f.lift_del_off = l_del_off.data(); f.lift_del = l_del.data();
        }
        return moni_index_create(&f, device, out);
    } catch (const std::bad_alloc&) {
        return MONI_ENOMEM;
    }
}

void moni_index_destroy(moni_index_t* I) {
    if (!I) return;
    (void)hipSetDevice(I->device);
    void* ps[] = {I->d_tables, I->d_rows, I->d_frows, I->d_cr, I->d_recs, I->d_phi, I->d_phi_inv, I->d_phi_dir, I->d_phi_inv_dir, I->d_text, I->d_text2, I->d_exc, I->d_seq_starts, I->d_name_id, I->d_snames, I->d_sname_off, I->d_lift_seqs, I->d_lift_runs, I->d_pdir};
    for (void* p : ps) if (p) (void)hipFree(p);
    delete I;
}
uint64_t moni_index_n(const moni_index_t* I) { return I ? I->K.n : 0; }
uint64_t moni_index_r(const moni_index_t* I) { return I ? I->K.r : 0; }
uint64_t moni_index_device_bytes(const moni_index_t* I) { return I ? I->bytes : 0; }
int moni_index_text(const moni_index_t* I, uint8_t* out, uint64_t cap) {
    if (!I || !out || cap < I->h_text.size()) return MONI_EINVAL;
    memcpy(out, I->h_text.data(), I->h_text.size());
    return MONI_OK;
}

int moni_ctx_create(moni_index_t* I, moni_ctx_t** out) {
    if (!I || !out) return MONI_EINVAL;
    HIPCHK(hipSetDevice(I->device));
    moni_ctx* c = new moni_ctx();
    c->idx = I;
    if (const char* v = getenv("MONI_MS_VARIANT")) c->ms_variant = atoi(v);
    if (const char* v = getenv("MONI_EXTZ_LDS")) c->extz_lds = atoi(v);
    for (int i = 0; i < EV_N; ++i) { c->ev[i] = nullptr; c->ev_valid[i] = false; }
    bool ok = hipStreamCreate(&c->stream) == hipSuccess;
    for (int i = 0; ok && i < EV_N; ++i) ok = hipEventCreate(&c->ev[i]) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_small, 16) == hipSuccess && hipMalloc((void**)&c->d_counters, 4 * sizeof(unsigned long long)) == hipSuccess &&
         hipMemset(c->d_small, 0, 16) == hipSuccess && hipMemset(c->d_counters, 0, 4 * sizeof(unsigned long long)) == hipSuccess;
    if (!ok) { fprintf(stderr, "moni_hip: context creation failed on device %d\n", I->device); moni_ctx_destroy(c); return MONI_ENODEV; }
    *out = c;
    return MONI_OK;
}

void moni_ctx_destroy(moni_ctx_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->idx->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& x : c->stash) { x.seq.release(); x.offs.release(); x.blk.release(); }
    c->blk.release(); c->seq.release(); c->offs.release(); c->ptr.release(); c->pat.release(); c->pflag.release(); c->cnt_m.release(); c->cnt_s.release(); c->mem_slots.release(); c->tot.release();
    c->read_mem_off.release(); c->mems.release(); c->aux.release(); c->lowers.release(); c->tmp.release();
    c->occ_cnt.release(); c->occ_off.release(); c->occs.release(); c->pool.release(); c->scan_tmp.release();
    for (int x = 0; x < AK_NSET; ++x) c->af[x].release();
    for (int x = 0; x < PE_NSET; ++x) c->af_pe[x].release();
    for (int x = 0; x < PE_NSTREAM; ++x) if (c->pe_stream[x]) (void)hipStreamDestroy(c->pe_stream[x]);
    c->af_ctr_host.release(); c->pe_hcur.release(); c->fb_all.release();
    c->dp_q.release(); c->dp_t.release(); c->dp_dir.release(); c->dp_tasks.release(); c->dp_res.release(); c->dp_cig.release();
    c->dp_off.release(); c->dp_ws.release(); c->dp_big.release(); c->dp_dir_big.release(); c->ak_slots.release(); c->ak_waves.release(); c->ak_cursors.release(); c->ak_rnames.release(); c->ak_quals.release(); c->ak_rname_off.release(); c->ak_txt.release(); c->ak_mapq_tab.release(); c->h_txt.release(); c->h_recs.release(); c->h_cig.release(); c->h_alt.release(); c->h_md.release();
    for (auto& ob : c->pieces) ob.release();
    for (int x = 0; x < AK_NSET; ++x) if (c->ak_stream[x]) (void)hipStreamDestroy(c->ak_stream[x]);
    for (int x = 0; x < AK_NSET; ++x) if (c->fb_stream[x]) (void)hipStreamDestroy(c->fb_stream[x]);
    for (auto e : c->ak_fin) (void)hipEventDestroy(e);
    for (auto e : c->af_ev) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (auto e : c->ak_begin) (void)hipEventDestroy(e);
    for (auto e : c->ak_done) (void)hipEventDestroy(e);
    c->ak_recs.release(); c->ak_cig.release(); c->ak_alt.release(); c->ak_minscore.release(); c->pe.release();
    if (c->d_ak_cursors) (void)hipFree(c->d_ak_cursors);
    if (c->out_buf) (void)hipHostFree(c->out_buf);
    for (int x = 0; x < AK_NSET; ++x) c->gather_tmp[x].release();
    c->ak_block.release(); c->ak_dev_len.release(); c->ak_dev_off.release(); c->ak_dev_pos.release(); c->ak_dev_sum.release(); c->h_sum.release();
    if (c->d_small) (void)hipFree(c->d_small);
    if (c->d_counters) (void)hipFree(c->d_counters);
    for (int i = 0; i < EV_N; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int reads_upload(moni_ctx* c, const moni_read_batch_t* b, bool keep_host_copy);
int moni_reads_upload(moni_ctx_t* c, const moni_read_batch_t* b) { return reads_upload(c, b, true); }

// keep_host_copy: the host pipeline (GpuBackend) reads the batch through c->h_seq / c->h_offs
static int reads_upload(moni_ctx* c, const moni_read_batch_t* b, bool keep_host_copy) {
    if (!c || !b || !b->offsets || (b->n_reads && !b->seq)) return MONI_EINVAL;
    HIPCHK(hipSetDevice(c->idx->device));
    const uint64_t nr = b->n_reads;
    uint64_t mx = 0;
    for (uint64_t i = 0; i < nr; ++i) {
        if (b->offsets[i + 1] < b->offsets[i]) return MONI_EINVAL;
        uint64_t l = b->offsets[i + 1] - b->offsets[i];
        if (l > mx) mx = l;
    }
    if (mx >= (1ull << 31)) return MONI_EINVAL;
    const uint64_t total = nr ? b->offsets[nr] - b->offsets[0] : 0;
    int rc;
    // workspace layout (seed_core.h): per block of 32 reads (64 tasks) as many steps as its longest read has
    const uint64_t n_blk = (nr + 31) / 32;
    std::vector<moni_u64x2> blk(n_blk + 1);
    {
        uint64_t pw = 0, qw = 0;
        for (uint64_t k = 0; k < n_blk; ++k) {
            uint64_t lb = 0;
            for (uint64_t i = 32 * k; i < nr && i < 32 * k + 32; ++i) lb = std::max<uint64_t>(lb, b->offsets[i + 1] - b->offsets[i]);
            blk[k].x = qw; blk[k].y = pw;
            qw += 64 * lb; pw += 64 * ws_pat_words(lb);
        }
        blk[n_blk].x = qw; blk[n_blk].y = pw;
    }
    if ((rc = c->seq.ensure(total + 16)) || (rc = c->offs.ensure(nr + 1)) || (rc = c->blk.ensure(n_blk + 1))) return rc;
    HIPCHK(hipMemcpyAsync(c->blk.p, blk.data(), (n_blk + 1) * sizeof(moni_u64x2), hipMemcpyHostToDevice, c->stream));
    std::vector<uint64_t> rel(nr + 1);
    for (uint64_t i = 0; i <= nr; ++i) rel[i] = b->offsets[i] - b->offsets[0];
    if (total) HIPCHK(hipMemcpyAsync(c->seq.p, b->seq + b->offsets[0], total, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->offs.p, rel.data(), (nr + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->n_reads = nr; c->total_len = total; c->max_len = mx;
    c->h_blk.swap(blk);
    if (keep_host_copy) { c->h_seq.assign(b->seq + b->offsets[0], b->seq + b->offsets[0] + total); c->h_offs = rel; }
    else { c->h_seq.clear(); c->h_offs.clear(); }
    c->n_mems = c->n_occs = 0;
    return MONI_OK;
}

int moni_reads_swap(moni_ctx_t* c, uint32_t slot) {
    if (!c || slot >= 256) return MONI_EINVAL;
    HIPCHK(hipSetDevice(c->idx->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    try { if (c->stash.size() <= slot) c->stash.resize(slot + 1); } catch (const std::bad_alloc&) { return MONI_ENOMEM; }
    moni_ctx::Stash& x = c->stash[slot];
    std::swap(c->seq, x.seq); std::swap(c->offs, x.offs); std::swap(c->blk, x.blk); c->h_blk.swap(x.h_blk); std::swap(c->n_reads, x.n_reads); std::swap(c->total_len, x.total_len); std::swap(c->max_len, x.max_len);
    c->h_seq.swap(x.h_seq); c->h_offs.swap(x.h_offs);
    c->n_mems = c->n_occs = 0;
    return MONI_OK;
}

static int ms_launch(moni_ctx* c) {
    moni_index* I = c->idx;
    const uint64_t n_tasks = 2 * c->n_reads;
    int rc;
    if (c->h_blk.empty()) return MONI_EINVAL;
    if ((rc = c->ptr.ensure(c->h_blk.back().x + 1)) || (rc = c->pat.ensure(c->h_blk.back().y + 1)) || (rc = c->pflag.ensure(n_tasks + 8))) return rc;
    const unsigned grid = (unsigned)((n_tasks + MS_BLOCK - 1) / MS_BLOCK);
    if (n_tasks)
        hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(MS_BLOCK), 0, c->stream, I->K, I->d_tables, c->seq.p, c->offs.p, c->blk.p, n_tasks, c->pat.p, c->pflag.p);
    rec(c, EV_MS0);
    if (n_tasks) {
#define MS_LAUNCH(NCH, MINW) do { const uint64_t nl = (n_tasks + (NCH) - 1) / (NCH); \
        hipLaunchKernelGGL((ms_lf_kernel<NCH, MINW>), dim3((unsigned)((nl + MS_BLOCK - 1) / MS_BLOCK)), dim3(MS_BLOCK), 0, c->stream, \
                           I->K, I->d_tables, I->d_rows, I->d_frows, I->d_cr, I->d_recs, c->pat.p, c->offs.p, c->blk.p, n_tasks, c->ptr.p, c->d_counters); } while (0)
        switch (c->ms_variant) {
            case 1: MS_LAUNCH(1, 8); break;
            case 2: MS_LAUNCH(2, 8); break;
            case 3: MS_LAUNCH(2, 6); break;
            case 4: MS_LAUNCH(1, 6); break;
            case 5: MS_LAUNCH(4, 4); break;
            default: MS_LAUNCH(1, 6); break;   // fastest in profiles/r01d_ms_variant_sweep_fastrows.txt
        }
#undef MS_LAUNCH
    }
    rec(c, EV_MS1);
    HIPCHK(hipGetLastError());
    return MONI_OK;
}

int moni_ms_run(moni_ctx_t* c) {
    if (!c) return MONI_EINVAL;
    HIPCHK(hipSetDevice(c->idx->device));
    HIPCHK(hipMemsetAsync(c->d_counters, 0, 4 * sizeof(unsigned long long), c->stream));
    rec(c, EV_ALL0);
    int rc = ms_launch(c);
    rec(c, EV_ALL1);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return MONI_OK;
}

// pointer of step s of a task, in the host copy of the pointer workspace (seed_core.h: ws_ptr_base)
static inline uint64_t ws_ptr_host(const moni_ctx* c, const std::vector<uint64_t>& h, uint64_t task, uint64_t s) { return h[c->h_blk[task >> 6].x + (task & 63u) + s * 64u]; }

int moni_ms_query_batch(moni_ctx_t* c, const moni_read_batch_t* b, uint64_t* pointers) {
    if (!pointers) return MONI_EINVAL;
    int rc = moni_reads_upload(c, b);
    if (rc) return rc;
    if ((rc = moni_ms_run(c))) return rc;
    try {
        std::vector<uint64_t> h(c->h_blk.back().x);
        if (!h.empty()) HIPCHK(hipMemcpy(h.data(), c->ptr.p, h.size() * 8, hipMemcpyDeviceToHost));
        const uint64_t base = b->offsets[0];
        for (uint64_t rd = 0; rd < c->n_reads; ++rd) {
            const uint64_t off = b->offsets[rd] - base, m = b->offsets[rd + 1] - b->offsets[rd];
            for (uint64_t s = 0; s < 2; ++s)
                for (uint64_t k = 0; k < m; ++k)
                    pointers[2 * off + s * m + k] = ws_ptr_host(c, h, 2 * rd + s, m - 1 - k);
        }
    } catch (const std::bad_alloc&) { return MONI_ENOMEM; }
    return MONI_OK;
}

// Legacy `moni ms` (src/matching_statistics.cpp:236-278): pointers and lengths of the forward strand of every read
int moni_ms_lengths_batch(moni_ctx_t* c, const moni_read_batch_t* b, uint64_t* pointers, uint64_t* lengths) {
    if (!pointers || !lengths) return MONI_EINVAL;
    int rc = moni_reads_upload(c, b);
    if (rc) return rc;
    if ((rc = moni_ms_run(c))) return rc;
    moni_index* I = c->idx;
    const uint64_t nr = c->n_reads;
    if (!nr) return MONI_OK;
    DBuf<uint32_t> lens;
    if ((rc = lens.ensure(c->total_len + 1))) return rc;
    hipLaunchKernelGGL(ms_len_kernel, dim3((unsigned)((nr + MS_BLOCK - 1) / MS_BLOCK)), dim3(MS_BLOCK), 0, c->stream, I->K, I->d_tables, I->d_text, c->pat.p, c->offs.p, c->blk.p, nr,
                       c->ptr.p, lens.p);
    try {
        std::vector<uint64_t> h(c->h_blk.back().x);
        std::vector<uint32_t> hl(c->total_len + 1);
        bool ok = hipStreamSynchronize(c->stream) == hipSuccess && (h.empty() || hipMemcpy(h.data(), c->ptr.p, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) &&
                  hipMemcpy(hl.data(), lens.p, c->total_len * 4, hipMemcpyDeviceToHost) == hipSuccess;
        lens.release();
        if (!ok) return MONI_ENODEV;
        const uint64_t base = b->offsets[0];
        for (uint64_t rd = 0; rd < nr; ++rd) {
            const uint64_t off = b->offsets[rd] - base, m = b->offsets[rd + 1] - b->offsets[rd];
            for (uint64_t k = 0; k < m; ++k) { pointers[off + k] = ws_ptr_host(c, h, 2 * rd, m - 1 - k); lengths[off + k] = hl[off + k]; }
        }
    } catch (const std::bad_alloc&) { lens.release(); return MONI_ENOMEM; }
    return MONI_OK;
}

// The seeding stage over the resident batch: MS pointers, MEMs (count, scan, emit), occurrences (count, scan, fill).
static int seed_all(moni_ctx* c, const moni_seed_params_t* prm) {
    moni_index* I = c->idx;
    const uint64_t nr = c->n_reads, n_tasks = 2 * nr;
    int rc;
    if ((rc = c->cnt_m.ensure(n_tasks + 2)) || (rc = c->cnt_s.ensure(n_tasks + 2)) || (rc = c->tot.ensure(nr + 2)) ||
        (rc = c->read_mem_off.ensure(nr + 2)) || (rc = c->mem_slots.ensure(n_tasks * MONI_MEM_SLOTS + 1)))
        return rc;
    uint32_t* cnt_m = c->cnt_m.p; uint32_t* cnt_s = c->cnt_s.p;
    uint64_t* tot = c->tot.p; uint64_t* rmo = c->read_mem_off.p;
    moni_u64x2* slots = c->mem_slots.p;
    const uint64_t* offs = c->offs.p;
    HIPCHK(hipMemsetAsync(c->d_counters, 0, 4 * sizeof(unsigned long long), c->stream));
    HIPCHK(hipMemsetAsync(c->d_small, 0, 16, c->stream));
    rec(c, EV_ALL0);
    if ((rc = ms_launch(c))) return rc;                              // (allocates pat / ptr)
    const uint64_t* pat = c->pat.p; const uint64_t* ptr = c->ptr.p;
    const unsigned grid_t = (unsigned)((n_tasks + MS_BLOCK - 1) / MS_BLOCK);
    const uint32_t split_on = prm->report_mems ? 0u : 1u;
    // the 2-bit comparison holds a lane's pattern in LDS: 5 words (160 bases) or 8 (256); reads beyond that compare bytes (MONI_MEM_BYTES=1: all do)
    static const bool mem_bytes = getenv("MONI_MEM_BYTES") != nullptr;
    const uint64_t* text2 = mem_bytes ? nullptr : I->d_text2;
#define MEM_LAUNCH(EMIT, RMO, MEMS, AUX) do { \
        if (c->max_len <= 160) hipLaunchKernelGGL((mem_kernel<EMIT, 5>), dim3(grid_t), dim3(MS_BLOCK), 0, c->stream, I->K, I->d_text, text2, I->d_exc, I->exc_sh, I->exc_words, pat, offs, c->blk.p, \
                                                  n_tasks, ptr, prm->min_len, split_on, cnt_m, cnt_s, RMO, MEMS, AUX, slots, c->d_counters); \
        else hipLaunchKernelGGL((mem_kernel<EMIT, 8>), dim3(grid_t), dim3(MS_BLOCK), 0, c->stream, I->K, I->d_text, text2, I->d_exc, I->exc_sh, I->exc_words, pat, offs, c->blk.p, \
                                n_tasks, ptr, prm->min_len, split_on, cnt_m, cnt_s, RMO, MEMS, AUX, slots, c->d_counters); } while (0)
    rec(c, EV_MC0);
    if (n_tasks)
        MEM_LAUNCH(false, (const uint64_t*)nullptr, (moni_mem_t*)nullptr, (uint32_t*)nullptr);
    rec(c, EV_MC1);
    hipLaunchKernelGGL(read_totals_kernel, dim3((unsigned)((nr + 1 + 255) / 256)), dim3(256), 0, c->stream, cnt_m, cnt_s, nr, tot);
    if ((rc = exclusive_scan_u64(c, tot, rmo, nr + 1))) return rc;
    uint64_t n_mems = 0;
    HIPCHK(hipMemcpyAsync(&n_mems, rmo + nr, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->tmp_cap = 16;
    {
        const uint64_t need = n_mems + 2;
        if ((rc = c->mems.ensure(need)) || (rc = c->aux.ensure(need)) || (rc = c->lowers.ensure(need)) || (rc = c->tmp.ensure(need * c->tmp_cap + 1)) ||
            (rc = c->occ_cnt.ensure(need + 1)) || (rc = c->occ_off.ensure(need + 1))) return rc;
        if ((rc = c->pool.ensure((size_t)c->pool_rows * I->K.n_seq + 1))) return rc;
    }
    moni_mem_t* mems = c->mems.p; uint32_t* aux = c->aux.p;
    uint64_t* occ_cnt = c->occ_cnt.p; uint64_t* occ_off = c->occ_off.p;
    rec(c, EV_ME0);
    if (n_tasks)
        MEM_LAUNCH(true, (const uint64_t*)rmo, mems, aux);
    rec(c, EV_ME1);
    occ_args_t A;
    A.phi.recs = I->d_phi; A.phi.dir = I->d_phi_dir; A.phi_inv.recs = I->d_phi_inv; A.phi_inv.dir = I->d_phi_inv_dir;
    A.text = I->d_text;
    A.seq_starts = I->d_seq_starts; A.name_id = I->d_name_id; A.mems = mems; A.aux = aux; A.read_mem_off = rmo;
    A.n_mems = n_mems; A.occs = nullptr; A.tmp = c->tmp.p; A.lowers = c->lowers.p; A.tmp_cap = c->tmp_cap;
    A.filter_seeds = prm->filter_seeds; A.n_seeds_thr = prm->n_seeds_thr; A.pool_rows = c->pool_rows; A.pool = c->pool.p;
    A.pool_next = c->d_small; A.error_flag = c->d_small + 1; A.counters = c->d_counters;
    const unsigned grid_m = (unsigned)((n_mems + MS_BLOCK - 1) / MS_BLOCK);
    // The per-genome filter needs a row of per-name counters only for seeds with more than n_seeds_thr occurrences;
    // rows come from a bump-allocated pool.  If a pass asks for more rows than the pool holds, the pool is grown to the
    // demand the pass reported and the pass is repeated (results of an exhausted pass are discarded).
    uint64_t n_occs = 0;
    uint32_t small[2] = {0, 0};
    for (int attempt = 0;; ++attempt) {
        A.pool = c->pool.p; A.pool_rows = c->pool_rows;
        HIPCHK(hipMemsetAsync(c->d_small, 0, 16, c->stream));
        HIPCHK(hipMemsetAsync(c->d_counters + 2, 0, sizeof(unsigned long long), c->stream));
        rec(c, EV_PC0);
        if (n_mems) hipLaunchKernelGGL(occ_kernel<false>, dim3(grid_m), dim3(MS_BLOCK), 0, c->stream, I->K, A);
        rec(c, EV_PC1);
        HIPCHK(hipMemcpyAsync(small, c->d_small, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (!small[1]) break;
        if (attempt >= 2) { fprintf(stderr, "moni_hip: per-genome counter pool exhausted (%u rows)\n", c->pool_rows); return MONI_ENOMEM; }
        c->pool_rows = small[0] + small[0] / 4 + 64;
        if ((rc = c->pool.ensure((size_t)c->pool_rows * I->K.n_seq + 1))) return rc;
    }
    hipLaunchKernelGGL(occ_cnt_gather_kernel, dim3((unsigned)((n_mems + 1 + 255) / 256)), dim3(256), 0, c->stream, mems, n_mems, occ_cnt);
    if ((rc = exclusive_scan_u64(c, occ_cnt, occ_off, n_mems + 1))) return rc;
    if (n_mems) hipLaunchKernelGGL(occ_off_scatter_kernel, dim3(grid_m), dim3(256), 0, c->stream, mems, n_mems, occ_off);
    HIPCHK(hipMemcpyAsync(&n_occs, occ_off + n_mems, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if ((rc = c->occs.ensure(n_occs + 1))) return rc;
    A.occs = c->occs.p;
    for (int attempt = 0;; ++attempt) {
        A.pool = c->pool.p; A.pool_rows = c->pool_rows;
        HIPCHK(hipMemsetAsync(c->d_small, 0, 16, c->stream));
        rec(c, EV_PE0);
        if (n_mems) hipLaunchKernelGGL(occ_kernel<true>, dim3(grid_m), dim3(MS_BLOCK), 0, c->stream, I->K, A);
        rec(c, EV_PE1);
        rec(c, EV_ALL1);
        HIPCHK(hipMemcpyAsync(small, c->d_small, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (!small[1]) break;
        if (attempt >= 2) { fprintf(stderr, "moni_hip: per-genome counter pool exhausted (%u rows)\n", c->pool_rows); return MONI_ENOMEM; }
        c->pool_rows = small[0] + small[0] / 4 + 64;
        if ((rc = c->pool.ensure((size_t)c->pool_rows * I->K.n_seq + 1))) return rc;
    }
    HIPCHK(hipGetLastError());
    c->n_mems = n_mems; c->n_occs = n_occs;
    return MONI_OK;
}

int moni_seed_run(moni_ctx_t* c, const moni_seed_params_t* prm) {
    if (!c || !prm) return MONI_EINVAL;
    HIPCHK(hipSetDevice(c->idx->device));
    return seed_all(c, prm);
}

int moni_seed_counts(moni_ctx_t* c, uint64_t* n_mems, uint64_t* n_occs) {
    if (!c) return MONI_EINVAL;
    if (n_mems) *n_mems = c->n_mems;
    if (n_occs) *n_occs = c->n_occs;
    return MONI_OK;
}

int moni_seed_fetch(moni_ctx_t* c, moni_mem_t* mems, uint64_t* occs, uint64_t* read_mem_off) {
    if (!c) return MONI_EINVAL;
    HIPCHK(hipSetDevice(c->idx->device));
    if (mems && c->n_mems) HIPCHK(hipMemcpy(mems, c->mems.p, c->n_mems * sizeof(moni_mem_t), hipMemcpyDeviceToHost));
    if (occs && c->n_occs) HIPCHK(hipMemcpy(occs, c->occs.p, c->n_occs * 8, hipMemcpyDeviceToHost));
    if (read_mem_off) HIPCHK(hipMemcpy(read_mem_off, c->read_mem_off.p, (c->n_reads + 1) * 8, hipMemcpyDeviceToHost));
    return MONI_OK;
}

int moni_seed_batch(moni_ctx_t* c, const moni_read_batch_t* b, const moni_seed_params_t* prm, moni_mem_t** mems, uint64_t* n_mems,
                    uint64_t** occs, uint64_t* n_occs, uint64_t** read_mem_off) {
    if (!mems || !n_mems || !occs || !n_occs || !read_mem_off) return MONI_EINVAL;
    int rc = moni_reads_upload(c, b);
    if (rc) return rc;
    if ((rc = moni_seed_run(c, prm))) return rc;
    *n_mems = c->n_mems; *n_occs = c->n_occs;
    *mems = (moni_mem_t*)malloc((c->n_mems + 1) * sizeof(moni_mem_t));
    *occs = (uint64_t*)malloc((c->n_occs + 1) * 8);
    *read_mem_off = (uint64_t*)malloc((c->n_reads + 1) * 8);
    if (!*mems || !*occs || !*read_mem_off) return MONI_ENOMEM;
    return moni_seed_fetch(c, *mems, *occs, *read_mem_off);
}

void moni_free(void* p) { free(p); }

int moni_phi_lcp_batch(moni_ctx_t* c, const uint64_t* pos, uint64_t n, int inverse, uint64_t* out_pos, uint64_t* out_lcp) {
    if (!c || (n && (!pos || !out_pos || !out_lcp))) return MONI_EINVAL;
    moni_index* I = c->idx;
    HIPCHK(hipSetDevice(I->device));
    if (!n) return MONI_OK;
    uint64_t* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, 3 * n * 8));
    if (hipMemcpy(d, pos, n * 8, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return MONI_ENODEV; }
    phi_tab_t P;
    P.recs = inverse ? I->d_phi_inv : I->d_phi;
    P.dir = inverse ? I->d_phi_inv_dir : I->d_phi_dir;
    hipLaunchKernelGGL(phi_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, I->K, P, d, n, d + n, d + 2 * n);
    const bool ok = hipStreamSynchronize(c->stream) == hipSuccess && hipMemcpy(out_pos, d + n, n * 8, hipMemcpyDeviceToHost) == hipSuccess &&
                    hipMemcpy(out_lcp, d + 2 * n, n * 8, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    return ok ? MONI_OK : MONI_ENODEV;
}

int moni_last_kernel_ms(moni_ctx_t* c, int which, float* ms) {
    if (!c || !ms || which < 0 || which > 6) return MONI_EINVAL;
    const int a = which == 6 ? EV_ALL0 : 2 * which, b = a + 1;
    if (!c->ev_valid[a] || !c->ev_valid[b]) return MONI_EINVAL;
    HIPCHK(hipEventSynchronize(c->ev[b]));
    HIPCHK(hipEventElapsedTime(ms, c->ev[a], c->ev[b]));
    return MONI_OK;
}

int moni_last_counters(moni_ctx_t* c, uint64_t out[4]) {
    if (!c || !out) return MONI_EINVAL;
    HIPCHK(hipSetDevice(c->idx->device));
    unsigned long long h[4];
    HIPCHK(hipMemcpy(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) out[i] = h[i];
    return MONI_OK;
}

#include "extz_host.inc"

}  // extern "C"

namespace {
// The product backend of the host pipeline: HIP kernels only.
struct GpuBackend : mh::Backend {
    moni_ctx* c;
    explicit GpuBackend(moni_ctx* c_) : c(c_) {}
    int seed(const moni_seed_params_t& p, std::vector<moni_mem_t>& mems, std::vector<uint64_t>& occs, std::vector<uint64_t>& rmo) override {
        int rc = moni_seed_run(c, &p);
        if (rc) return rc;
        mems.resize(c->n_mems); occs.resize(c->n_occs); rmo.resize(c->n_reads + 1);
        return moni_seed_fetch(c, mems.data(), occs.data(), rmo.data());
    }
    int dp(const moni_dp_params_t& p, const std::vector<moni_dp_task_t>& tasks, std::vector<moni_dp_result_t>& res, std::vector<uint32_t>& cig) override {
        res.resize(tasks.size());
        return dp_run(c, &p, nullptr, 0, nullptr, 0, tasks.data(), tasks.size(), res.data(), cig, true);
    }
};
}  // namespace

extern "C" {

// CPUs this process may actually use: affinity mask and cgroup v2 quota (a GPU box hands each job a CPU share)
static unsigned moni_host_cpus() {
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { unsigned a = (unsigned)CPU_COUNT(&set); if (a && a < n) n = a; }
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64]; unsigned long long period = 0;
        if (fscanf(f, "%63s %llu", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            unsigned c = (unsigned)((strtoull(q, nullptr, 10) + period - 1) / period);
            if (c && c < n) n = c;
        }
        fclose(f);
    }
    if (n < 1) n = 1;
    if (n > 128) n = 128;
    return n;
}

void moni_align_params_default(moni_align_params_t* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->min_len = 25; p->ext_len = 100; p->check_k = 5; p->region_dist = 10;
    p->filter_seeds = 1; p->n_seeds_thr = 1000; p->filter_freq = 1; p->left_mem_check = 1; p->freq_thr = 0.5;
    p->smatch = 2; p->smismatch = 4; p->gapo = 4; p->gapo2 = 13; p->gape = 2; p->gape2 = 1;
    p->end_bonus = 400; p->w = -1; p->zdrop = -1;
    p->max_dist_x = 500; p->max_dist_y = 100; p->max_iter = 10; p->max_pred = 5; p->min_chain_score = 40; p->min_chain_length = 1;
    p->host_threads = moni_host_cpus();
}

// The host pipeline (align_host.hpp) over a subset of the reads of the resident batch: used for reads the align kernel
// hands back (status 2) and as the whole path when MONI_ALIGN_HOST=1.
static int host_align_subset(moni_ctx* c, const moni_align_params_t& prm, const moni_read_batch_t& sub, const uint8_t* names,
                             const uint64_t* name_off, const uint8_t* quals, std::string& out, mh::AlignStats& st) {
    int rc = moni_reads_upload(c, &sub);
    if (rc) return rc;
    GpuBackend be(c);
    return mh::align_batch(be, c->idx->hix, prm, c->h_seq.data(), c->h_offs.data(), c->n_reads, names, name_off, quals, out, st);
}

static int align_core(moni_ctx* c, const moni_read_batch_t* b, bool resident, bool ctx_out, const uint8_t* names, const uint64_t* name_off, const uint8_t* quals,
                      const moni_align_params_t* prm, char** sam, uint64_t* sam_len, moni_align_stats_t* stats);

// `-c` (aligner_ksw2.hpp:340-343, 417; csv.hpp:55-67): the SAM records of the batch and one line of MEM statistics per read.  A diagnostics mode: the
// selection loop has to count the chains check_left_MEM skips, so every read goes through the host pipeline (align_host.hpp) over the GPU's seeds and DP
// batches; the per-genome occurrence counts of the seeds come from genome_kernel over the seeds of the last moni_seed_run.
static int genome_hi_lo(moni_ctx* c, const moni_align_params_t& prm, std::vector<uint64_t>& out) {
    moni_index* I = c->idx;
    const uint64_t n_mems = c->n_mems;
    out.assign(n_mems, 0);
    if (!n_mems) return MONI_OK;
    occ_args_t A;
    memset(&A, 0, sizeof A);
    A.text = I->d_text;
    A.phi.recs = I->d_phi; A.phi.dir = I->d_phi_dir; A.phi_inv.recs = I->d_phi_inv; A.phi_inv.dir = I->d_phi_inv_dir;
    A.seq_starts = I->d_seq_starts; A.name_id = I->d_name_id; A.mems = c->mems.p; A.aux = c->aux.p; A.read_mem_off = c->read_mem_off.p;
    A.n_mems = n_mems; A.lowers = c->lowers.p; A.filter_seeds = prm.filter_seeds; A.n_seeds_thr = prm.n_seeds_thr;
    const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / (4ull * I->K.n_seq));          // rows of per-name counters: 256 MB at a time
    DBuf<uint32_t> rows; DBuf<uint64_t> hl;
    int rc;
    if ((rc = rows.ensure(std::min(chunk, n_mems) * I->K.n_seq + 1)) || (rc = hl.ensure(n_mems + 1))) { rows.release(); hl.release(); return rc; }
    bool ok = true;
    for (uint64_t g0 = 0; g0 < n_mems && ok; g0 += chunk) {
        const uint64_t g1 = std::min(n_mems, g0 + chunk);
        ok = hipMemsetAsync(rows.p, 0, (g1 - g0) * I->K.n_seq * sizeof(uint32_t), c->stream) == hipSuccess;
        if (ok) hipLaunchKernelGGL(genome_kernel, dim3((unsigned)((g1 - g0 + MS_BLOCK - 1) / MS_BLOCK)), dim3(MS_BLOCK), 0, c->stream, I->K, A, g0, g1, rows.p, hl.p);
    }
    ok = ok && hipStreamSynchronize(c->stream) == hipSuccess && hipMemcpy(out.data(), hl.p, n_mems * 8, hipMemcpyDeviceToHost) == hipSuccess;
    rows.release(); hl.release();
    return ok ? MONI_OK : MONI_ENODEV;
}

int moni_align_csv_batch(moni_ctx_t* c, const moni_read_batch_t* b, const uint8_t* names, const uint64_t* name_off, const uint8_t* quals, const moni_align_params_t* prm,
                         char** sam, uint64_t* sam_len, char** csv, uint64_t* csv_len, moni_align_stats_t* stats) {
    if (!c || !b || !prm || !sam || !sam_len || !csv || !csv_len || (b->n_reads && (!names || !name_off))) return MONI_EINVAL;
    if (prm->w >= 0 || prm->zdrop >= 0) return MONI_EINVAL;
    HIPCHK(hipSetDevice(c->idx->device));
    try {
        int rc = moni_reads_upload(c, b);
        if (rc) return rc;
        GpuBackend be(c);
        std::string out, lines;
        mh::AlignStats st;
        const std::function<int(std::vector<uint64_t>&)> hl = [&](std::vector<uint64_t>& v) { return genome_hi_lo(c, *prm, v); };
        if ((rc = mh::align_batch(be, c->idx->hix, *prm, c->h_seq.data(), c->h_offs.data(), c->n_reads, names, name_off, quals ? quals + b->offsets[0] : nullptr, out, st, &hl, &lines))) return rc;
        char* a = (char*)malloc(out.size() + 1); char* d = (char*)malloc(lines.size() + 1);
        if (!a || !d) { free(a); free(d); return MONI_ENOMEM; }
        memcpy(a, out.data(), out.size()); a[out.size()] = 0; memcpy(d, lines.data(), lines.size()); d[lines.size()] = 0;
        *sam = a; *sam_len = out.size(); *csv = d; *csv_len = lines.size();
        if (stats) { memset(stats, 0, sizeof *stats); stats->reads = st.reads; stats->aligned = st.aligned; stats->dp_tasks = st.dp_tasks; stats->dp_cells = st.dp_cells; stats->dp_rounds = st.dp_rounds;
                     stats->t_seed = st.t_seed; stats->t_chain = st.t_chain; stats->t_dp = st.t_dp; stats->t_host = st.t_host; stats->handed_back = st.reads; }
        return MONI_OK;
    } catch (const std::bad_alloc&) { return MONI_ENOMEM; }
}

// Reads of MONI_LONG_READ bases or more do not go through the batch with the others: the seeding workspace is strided by the longest
// read of a batch, so one long read among 262144 short ones would multiply its size (a 100 kb read: hundreds of GB).  They are
// aligned as a batch of their own (host pipeline over the same kernels) and spliced back in; a read beyond the largest DP the
// kernels take (8192 bases) is reported unaligned, with a notice, instead of failing the batch.
#define MONI_LONG_READ 4096u
#define MONI_MAX_READ 8192u

int moni_align_batch(moni_ctx_t* c, const moni_read_batch_t* b, const uint8_t* names, const uint64_t* name_off, const uint8_t* quals,
                     const moni_align_params_t* prm, char** sam, uint64_t* sam_len, moni_align_stats_t* stats) {
    if (!c || !b || !prm || !sam || !sam_len || (b->n_reads && (!names || !name_off))) return MONI_EINVAL;
    const uint64_t NR = b->n_reads;
    std::vector<uint32_t> longs;
    for (uint64_t r = 0; r < NR; ++r) if (b->offsets[r + 1] - b->offsets[r] >= MONI_LONG_READ) longs.push_back((uint32_t)r);
    if (longs.empty()) return align_core(c, b, false, false, names, name_off, quals, prm, sam, sam_len, stats);
    try {
        // the short reads as a batch of their own
        std::vector<uint8_t> sseq, snm, sq; std::vector<uint64_t> soff(1, 0), snoff(1, 0);
        std::vector<uint8_t> lseq, lnm, lq; std::vector<uint64_t> loff(1, 0), lnoff(1, 0);
        std::vector<std::string> long_line(longs.size());
        size_t li = 0;
        for (uint64_t r = 0; r < NR; ++r) {
            const uint64_t off = b->offsets[r], m = b->offsets[r + 1] - off;
            const bool is_long = li < longs.size() && longs[li] == r;
            if (is_long && m > MONI_MAX_READ) {        // no kernel takes it: unaligned record (sam.hpp:144-188, flag 4)
                fprintf(stderr, "moni_hip: read %llu has %llu bases (more than %u): reported unaligned\n", (unsigned long long)r, (unsigned long long)m, MONI_MAX_READ);
                std::string& ln = long_line[li];
                ln.assign((const char*)names + name_off[r], (const char*)names + name_off[r + 1]);
                ln += "\t4\t*\t0\t255\t*\t*\t0\t0\t";
                ln.append((const char*)b->seq + off, m); ln.push_back('\t');
                if (quals) ln.append((const char*)quals + off, m); else ln.push_back('*');
                ln.push_back('\n');
                ++li;
                continue;
            }
            std::vector<uint8_t>& S = is_long ? lseq : sseq; std::vector<uint8_t>& N = is_long ? lnm : snm; std::vector<uint8_t>& Q = is_long ? lq : sq;
            std::vector<uint64_t>& O = is_long ? loff : soff; std::vector<uint64_t>& NO = is_long ? lnoff : snoff;
            S.insert(S.end(), b->seq + off, b->seq + off + m); O.push_back(S.size());
            N.insert(N.end(), names + name_off[r], names + name_off[r + 1]); NO.push_back(N.size());
            if (quals) Q.insert(Q.end(), quals + off, quals + off + m);
            if (is_long) ++li;
        }
        char* ssam = nullptr; uint64_t slen = 0;
        moni_align_stats_t st_s; memset(&st_s, 0, sizeof st_s);
        const moni_read_batch_t sb{sseq.data(), soff.data(), (uint64_t)soff.size() - 1};
        int rc = align_core(c, &sb, false, false, snm.data(), snoff.data(), quals ? sq.data() : nullptr, prm, &ssam, &slen, &st_s);
        if (rc) return rc;
        std::string lsam; mh::AlignStats st_l;
        if (loff.size() > 1) {
            const moni_read_batch_t lb{lseq.data(), loff.data(), (uint64_t)loff.size() - 1};
            if ((rc = host_align_subset(c, *prm, lb, lnm.data(), lnoff.data(), quals ? lq.data() : nullptr, lsam, st_l))) { free(ssam); return rc; }
        }
        // splice in input order
        std::string out;
        out.reserve(slen + lsam.size() + 64 * longs.size());
        size_t ps = 0, pl = 0; li = 0;
        for (uint64_t r = 0; r < NR; ++r) {
            if (li < longs.size() && longs[li] == r) {
                if (!long_line[li].empty()) out += long_line[li];
                else { const size_t e = lsam.find('\n', pl); out.append(lsam, pl, e - pl + 1); pl = e + 1; }
                ++li;
            } else { const char* e = (const char*)memchr(ssam + ps, '\n', slen - ps); const size_t n = (size_t)(e - (ssam + ps)) + 1; out.append(ssam + ps, n); ps += n; }
        }
        free(ssam);
        char* dst = (char*)malloc(out.size() + 1);
        if (!dst) return MONI_ENOMEM;
        memcpy(dst, out.data(), out.size()); dst[out.size()] = 0;
        *sam = dst; *sam_len = out.size();
        if (stats) { *stats = st_s; stats->reads = NR; stats->aligned += st_l.aligned; stats->dp_tasks += st_l.dp_tasks; stats->dp_cells += st_l.dp_cells; stats->handed_back += (uint64_t)loff.size() - 1; }
        return MONI_OK;
    } catch (const std::bad_alloc&) { return MONI_ENOMEM; }
}

int moni_align_stream(moni_ctx_t* c, const moni_read_batch_t* b, const uint8_t* names, const uint64_t* name_off, const uint8_t* quals,
                      const moni_align_params_t* prm, char** sam, uint64_t* sam_len, moni_align_stats_t* stats) {
    if (!c || !b || !prm || !sam || !sam_len || (b->n_reads && (!names || !name_off))) return MONI_EINVAL;
    bool any_long = false;
    for (uint64_t r = 0; r < b->n_reads && !any_long; ++r) any_long = b->offsets[r + 1] - b->offsets[r] >= MONI_LONG_READ;
    if (!any_long) return align_core(c, b, false, true, names, name_off, quals, prm, sam, sam_len, stats);
    // long reads in the batch (rare): the splicing path of moni_align_batch, its text moved into the context's buffer
    char* tmp = nullptr; uint64_t len = 0;
    int rc = moni_align_batch(c, b, names, name_off, quals, prm, &tmp, &len, stats);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->idx->device));
    if (c->out_cap < len + 1) {
        char* nb = nullptr;
        if (hipHostMalloc((void**)&nb, len + 1, hipHostMallocDefault) != hipSuccess) { free(tmp); return MONI_ENOMEM; }
        if (c->out_buf) (void)hipHostFree(c->out_buf);
        c->out_buf = nb; c->out_cap = len + 1;
    }
    memcpy(c->out_buf, tmp, len); c->out_buf[len] = 0;
    free(tmp);
    *sam = c->out_buf; *sam_len = len;
    return MONI_OK;
}

int moni_align_run(moni_ctx_t* c, const uint8_t* names, const uint64_t* name_off, const uint8_t* quals, const moni_align_params_t* prm,
                   char** sam, uint64_t* sam_len, moni_align_stats_t* stats) {
    if (!c || !prm || !sam || !sam_len || (c->n_reads && (!names || !name_off))) return MONI_EINVAL;
    if (c->h_offs.size() != c->n_reads + 1) return MONI_EINVAL;          // no batch resident (moni_reads_upload first)
    // the host copy of the resident batch is the view the host stage formats SEQ from; it is taken out of the context for
    // the duration of the call because the hand-back path makes its own (small) batch resident
    std::vector<uint8_t> hs = std::move(c->h_seq);
    std::vector<uint64_t> ho = std::move(c->h_offs);
    const moni_read_batch_t view{hs.data(), ho.data(), c->n_reads};
    const uint64_t nr = c->n_reads, tl = c->total_len, ml = c->max_len;
    int rc = align_core(c, &view, true, true, names, name_off, quals, prm, sam, sam_len, stats);
    if (c->n_reads != nr || c->h_offs.size() != 0) {                     // the hand-back path replaced the resident batch: put it back
        const int rc2 = reads_upload(c, &view, false);
        if (!rc) rc = rc2;
    }
    c->h_seq = std::move(hs); c->h_offs = std::move(ho);
    c->n_reads = nr; c->total_len = tl; c->max_len = ml;
    return rc;
}

static int align_core(moni_ctx* c, const moni_read_batch_t* b, bool resident, bool ctx_out, const uint8_t* names, const uint64_t* name_off, const uint8_t* quals,
                      const moni_align_params_t* prm, char** sam, uint64_t* sam_len, moni_align_stats_t* stats) {
    moni_index* I = c->idx;
    HIPCHK(hipSetDevice(I->device));          // a worker thread of moni-hip-align starts on device 0: every allocation below must land on this index's GPU
    c->dp_kernel_ms_accum = 0;
    std::string out;
    mh::AlignStats st;
    uint64_t why_out[12] = {};
    const uint8_t* q = quals ? quals + b->offsets[0] : nullptr;
    static const bool host_only = getenv("MONI_ALIGN_HOST") != nullptr;
    bool out_done = false;
    int rc;
    if (host_only) {
        if ((rc = host_align_subset(c, *prm, *b, names, name_off, q, out, st))) return rc;
    } else {
        // ---- seeds stay in HBM; align_kernel takes every read from seeds to a finished alignment record.  The batch goes
        // through in sub-batches whose launches are all queued up front, alternating between two streams (each with its
        // own set of in-flight read slots) so that the next launch fills the CUs the previous one's stragglers leave;
        // a host thread follows behind: as a sub-batch's launch completes it fetches the records and the host threads turn
        // them into SAM text (MD/NM, MAPQ, formatting) while the GPU works on the later sub-batches ----
        if (prm->w >= 0 || prm->zdrop >= 0) return MONI_EINVAL;
        const double t_enter = mh::now_s();
        double t_mark[4] = {0, 0, 0, 0};
        const uint64_t NR = b->n_reads;
        // sub-batch schedule: a small first launch (the host stage starts after it), larger ones in the middle (a launch wants
        // several reads per lane in flight: ~4000 waves x AK_NL lanes), a small last one (the host stage left over after it is short)
        uint64_t sub_min = 250000, sub_mid = 250000;      // ~one read per lane and launch (4096 waves x AK_NL = 64 lanes); equal pieces: measured, profiles/sweep_align_nl.sh
        if (const char* v = getenv("MONI_ALIGN_SUB")) { const long long x = atoll(v); if (x > 0) sub_min = sub_mid = (uint64_t)x; }
        if (const char* v = getenv("MONI_ALIGN_SUB_MIN")) { const long long x = atoll(v); if (x > 0) sub_min = (uint64_t)x; }
        std::vector<uint64_t> sub_lo(1, 0);
        while (sub_lo.back() < NR) {
            const uint64_t done = sub_lo.back(), rest = NR - done;
            uint64_t take = done == 0 ? sub_min : (rest >= sub_mid + sub_min ? sub_mid : (rest > sub_min + sub_min / 2 ? rest - sub_min : rest));
            if (take > rest) take = rest;
            sub_lo.push_back(done + take);
        }
        const uint64_t n_sub = sub_lo.size() - 1;
        uint64_t sub_reads = 0;                                   // the largest sub-batch (pool shares are sized for it)
        for (uint64_t k = 0; k < n_sub; ++k) sub_reads = std::max(sub_reads, sub_lo[k + 1] - sub_lo[k]);
        uint64_t force_back = 0;        // test hook: treat every n-th read as handed back by the kernel (exercises that path)
        if (const char* v = getenv("MONI_AK_FORCE_HANDBACK")) { const long long x = atoll(v); if (x > 0) force_back = (uint64_t)x; }
        if (!c->ak_waves_full) {      // persistent waves: exactly as many blocks as stay resident, each lane takes reads off a shared counter
            int n_cu = 256, per_cu = 8;
            n_cu = ctx_n_cu(c);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, align_kernel, 64, 0) != hipSuccess || per_cu < 1) per_cu = 8;
            c->ak_waves_full = (uint64_t)n_cu * (uint64_t)per_cu;
        }
        const uint64_t waves_full = c->ak_waves_full;
        const int T = prm->host_threads > 0 ? (int)prm->host_threads : 1;
        mh::Pool pool(T);
        mh::Aligner AL(I->hix, *prm, b->seq, b->offsets);
        if ((int)c->pieces.size() < T) c->pieces.resize(T);
        std::vector<std::vector<std::string>> kept(n_sub);           // text of sub-batches that could not be assembled eagerly (hand-back path)
        std::vector<std::vector<uint32_t>> back_of(n_sub);
        std::vector<uint64_t> aligned_t(T, 0);
        double host_busy = 0;
        // the batch's text is assembled as the sub-batches finish (inside the overlapped host stage), as long as no read
        // has been handed back
        char* abuf = ctx_out ? c->out_buf : nullptr; size_t acap = ctx_out ? c->out_cap : 0, alen = 0; uint64_t eager_upto = 0; bool eager_ok = true, eager_oom = false;
        auto drop_abuf = [&]() { if (!ctx_out) free(abuf); abuf = nullptr; };
        auto grow_abuf = [&](size_t cap) -> bool {        // keeps the first alen bytes
            if (!ctx_out) { char* nb = (char*)realloc(abuf, cap); if (!nb) return false; abuf = nb; acap = cap; return true; }
            char* nb = nullptr;
            if (hipHostMalloc((void**)&nb, cap, hipHostMallocDefault) != hipSuccess) return false;
            if (abuf && alen) memcpy(nb, abuf, alen);
            if (abuf) (void)hipHostFree(abuf);
            abuf = nb; acap = cap; c->out_buf = nb; c->out_cap = cap;
            return true;
        };
        double prof[7] = {0, 0, 0, 0, 0, 0, 0};
        uint64_t why_sum[AF_WHY_N] = {};
        double hist[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cyc[7] = {0, 0, 0, 0, 0, 0, 0};
        uint64_t waves_used = 0;

        // SAM text in the kernel (default; MONI_ALIGN_HOST_FORMAT=1 leaves the formatting to the host stage): read names and
        // qualities go to the device while the seeding runs
        const bool gpu_text = getenv("MONI_ALIGN_HOST_FORMAT") == nullptr && NR > 0;
        const bool fin_v1 = getenv("MONI_AF_FIN_V1") != nullptr;          // finish_wave_kernel (one wavefront per read for all of it) instead of finish_prep_kernel + finish_render_kernel
        std::thread uploader;
        int rc_up = MONI_OK;
        if (gpu_text) {
            const uint64_t nb_names = name_off[NR] - name_off[0], nb_q = quals ? b->offsets[NR] - b->offsets[0] : 0;
            if ((rc = c->ak_rnames.ensure(nb_names + 16)) || (rc = c->ak_rname_off.ensure(NR + 1)) || (rc = c->ak_quals.ensure(nb_q + 16))) return rc;
            if (!c->copy_stream) HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
            uploader = std::thread([&, nb_names, nb_q]() {
                if (hipSetDevice(I->device) != hipSuccess) { rc_up = MONI_ENODEV; return; }
                std::vector<uint64_t> rel(NR + 1);
                for (uint64_t r = 0; r <= NR; ++r) rel[r] = name_off[r] - name_off[0];
                if (hipMemcpyAsync(c->ak_rnames.p, names + name_off[0], nb_names, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
                    hipMemcpyAsync(c->ak_rname_off.p, rel.data(), (NR + 1) * 8, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
                    (nb_q && hipMemcpyAsync(c->ak_quals.p, quals + b->offsets[0], nb_q, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess) ||
                    hipStreamSynchronize(c->copy_stream) != hipSuccess) rc_up = MONI_ENODEV;
            });
        }
        struct JoinGuard { std::thread& t; ~JoinGuard() { if (t.joinable()) t.join(); } } join_guard{uploader};
        // The staged kernels (align_fast.hip) take the common case; align_kernel takes the reads they hand over (MONI_ALIGN_V1=1: every read)
        static const bool use_fast = getenv("MONI_ALIGN_V1") == nullptr;
        // seeding runs over the whole batch before the align kernels of its sub-batches (seeding sub-batch k+1 beside the align kernels of
        // sub-batch k was measured in round 2 and did not pay inside one context: the two align streams already fill the GPU; a streaming
        // caller gets that overlap from two or three contexts per GPU)
        moni_seed_params_t sp;
        sp.min_len = prm->min_len; sp.filter_seeds = prm->filter_seeds; sp.n_seeds_thr = prm->n_seeds_thr; sp.report_mems = 0;
        {
            const double t0 = mh::now_s();
            if (!resident && (rc = reads_upload(c, b, false))) return rc;
            if ((rc = moni_seed_run(c, &sp))) return rc;
            st.t_seed += mh::now_s() - t0;
        }
        t_mark[0] = mh::now_s() - t_enter;
        const double t_gpu0 = mh::now_s();
        double t_launch[3] = {0, 0, 0};
        // device side of the align stage
        std::vector<int32_t> msc(c->max_len + 2);
        for (uint64_t l = 0; l <= c->max_len + 1; ++l) msc[l] = l ? (int32_t)(20 + 8 * log((double)l)) : INT32_MIN;   // aligner_ksw2.hpp:394
        // records, CIGARs and alternative hits are written by the kernel straight into pinned host memory (a few hundred
        // bytes per read over PCIe): no copy has to find room next to the persistent kernels.  Pool share of one sub-batch
        // below; a launch that runs out hands the affected reads back (status 2)
        const uint64_t cig_per = 16 * sub_reads + 4096, alt_per = 24 * sub_reads + 4096, md_per = 8 * sub_reads + 4096;      // md: 8-byte words
        const uint64_t txt_per = gpu_text ? 160 * sub_reads + 4096 : 0;                        // text: 8-byte words (1.25 KB per read on average)
        if (uploader.joinable()) uploader.join();
        if (rc_up) return rc_up;
        if (gpu_text) {
            const uint32_t TAB = 8192;
            if ((rc = c->ak_txt.ensure(txt_per * n_sub + 8)) || (rc = c->h_txt.ensure(txt_per + 8)) || (rc = c->ak_mapq_tab.ensure(TAB))) return rc;
            if (!c->mapq_tab_ready) {
                std::vector<double> tab(TAB, 1.);
                const int32_t coeff_fac = (int32_t)log(50.0f);                                 // aligner_ksw2.hpp:3250-3251
                for (int32_t l = 2; l < (int32_t)TAB; ++l) tab[l] = coeff_fac / log(l);         // the expression of mapq.hpp:170, libm on the host
                HIPCHK(hipMemcpy(c->ak_mapq_tab.p, tab.data(), TAB * sizeof(double), hipMemcpyHostToDevice));
                c->mapq_tab_ready = true;
            }
        }
        int n_cu = 256;
        n_cu = ctx_n_cu(c);
        const uint64_t ak_waves = use_fast ? std::min<uint64_t>(waves_full, 1024) : waves_full;       // in-flight read slots of align_kernel (75 KB each)
        if ((rc = c->ak_slots.ensure(AK_NSET * ak_waves * AK_NL)) || (rc = c->ak_waves.ensure(AK_NSET * ak_waves)) || (rc = c->h_recs.ensure(NR + 1)) ||
            (rc = c->h_cig.ensure(cig_per * n_sub + 1)) || (rc = c->h_alt.ensure(alt_per * n_sub + 1)) || (rc = c->h_md.ensure(md_per * n_sub + 1)) || (rc = c->ak_minscore.ensure(msc.size())) ||
            (rc = c->ak_cursors.ensure(AK_CUR * n_sub + AK_CUR)))
            return rc;
        // task slots: AF_MAX_TASKS_READ per read (each read's problems at its own place) + the global problems; bin queues: as many entries as problems are expected
        const uint32_t af_slot_cap = (uint32_t)std::min<uint64_t>((uint64_t)AF_MAX_TASKS_READ * sub_reads + 4 * sub_reads + 4096, 0xFFFFFFF0ull);
        const uint32_t af_task_cap = (uint32_t)std::min<uint64_t>(12 * sub_reads + 4096, (1ull << AF_POS_BITS) - 1), af_tb_cap = (uint32_t)std::min<uint64_t>((4 + c->max_len / 50) * sub_reads + 1024, 0x7FFFFFFFull);      // traced problems: the final chain's extensions and gap fills (more anchors on longer reads)
        const uint32_t af_chunk_cap = af_task_cap / 64 + 2 * AF_NBIN;
        // direction bits: half a byte per DP cell.  ~10 KB per 150 bp read on the bench; DP cells grow with the square of the read length (250 bp:
        // ~90 KB with the global realignments); a chunk that does not fit sends its reads to align_kernel, and a batch in which that happened
        // doubles the budget of the next ones
        uint64_t dirs_per_read = 16384;          // (measured: 2.9 KB per 150 bp read with the banded problems, 11 KB with every problem on the full tiles, 21 KB for 250 bp reads on 20 haplotypes: profiles/r05a)
        { const double f = (double)c->max_len / 150.0; if (f > 1.0) dirs_per_read = (uint64_t)(16384.0 * f * f); }
        dirs_per_read = std::min<uint64_t>(dirs_per_read * c->dirs_scale, 1ull << 20);
        const uint64_t af_dirs_cap = dirs_per_read * sub_reads + (16ull << 20);
        const unsigned af_dp_grid = (unsigned)n_cu * 12;
        const unsigned af_fin_grid = (unsigned)std::min<uint64_t>((sub_reads + 63) / 64, (uint64_t)n_cu * 16);
        // in-order text on the GPU: when the caller takes the context-owned buffer and the kernels spell the text, every sub-batch's lines are
        // put in read order on the device and arrive with one transfer; the host threads are needed only for sub-batches with hand-backs
        const bool inorder = use_fast && gpu_text && ctx_out && force_back == 0 && getenv("MONI_ALIGN_HOST_ORDER") == nullptr;
        if (inorder && ((rc = c->ak_recs.ensure(NR + 1)) || (rc = c->ak_block.ensure(txt_per * n_sub + 8)) || (rc = c->ak_dev_len.ensure(NR + n_sub + 8)) || (rc = c->ak_dev_off.ensure(NR + n_sub + 8)) ||
                        (rc = c->ak_dev_pos.ensure(NR + 2 * n_sub + 8)) || (rc = c->ak_dev_sum.ensure(160 * n_sub + 8)) || (rc = c->h_sum.ensure(4 * n_sub + 4)))) return rc;
        size_t gather_tmp_bytes = 0;
        if (inorder) {
            if (rocprim::exclusive_scan(nullptr, gather_tmp_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, (size_t)sub_reads, rocprim::plus<uint64_t>(), c->stream) != hipSuccess) return MONI_ENODEV;
            for (int x = 0; x < AK_NSET; ++x) if ((rc = c->gather_tmp[x].ensure(gather_tmp_bytes + 16))) return rc;
        }
        if (inorder) { HIPCHK(hipMemsetAsync(c->ak_dev_sum.p, 0, (160 * n_sub + 8) * sizeof(unsigned long long), c->stream)); memset(c->h_sum.p, 0, (4 * n_sub + 4) * sizeof(unsigned long long)); }
        if (use_fast) for (int x = 0; x < (int)std::min<uint64_t>(n_sub, AK_NSET); ++x) {
            moni_ctx::AfSet& S = c->af[x];
            if ((rc = S.plans.ensure(sub_reads + 1)) || (rc = S.tasks.ensure(af_slot_cap)) || (rc = S.res.ensure(af_slot_cap)) || (rc = S.ntasks.ensure(sub_reads + 8)) || (rc = S.bin_q.ensure((size_t)(AF_NBIN + 1) * af_task_cap)) ||
                (rc = S.task_pos.ensure(af_slot_cap)) || (rc = S.tb_task.ensure(af_tb_cap)) || (rc = S.tb.ensure(af_tb_cap)) || (rc = S.big_list.ensure(3 * (sub_reads + 1))) || (rc = S.ctab.ensure((size_t)(sub_reads + 1) * AF_CTAB)) ||
                (rc = S.ctr.ensure(AF_NCTR)) || (rc = S.txt_cur.ensure(AF_TXT_SHARDS * 8)) || (rc = S.bnd.ensure((size_t)af_dp_grid * AF_QCAP * 64)) || (rc = S.chunks.ensure(af_chunk_cap)) || (rc = S.dirs.ensure(af_dirs_cap)) || (rc = S.fin.ensure((size_t)af_fin_grid * 64 * sizeof(af_fin_t))) || (gpu_text && !fin_v1 && (rc = S.recipes.ensure((size_t)(sub_reads + 1) * AFP_WORDS))))
                return rc;
        }
        // the launches alternate between the context's stream and one more: HIP multiplexes streams onto a handful of hardware queues
        // (4 by default), and two streams that land on the same queue run their kernels one after the other
        for (int x = 0; x < AK_NSET; ++x) if (!c->ak_stream[x]) HIPCHK(hipStreamCreateWithFlags(&c->ak_stream[x], hipStreamNonBlocking));
        if (!c->copy_stream) HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        while (c->ak_done.size() < n_sub) { hipEvent_t e0, e1, e2; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventCreate(&e2));
                                            c->ak_begin.push_back(e0); c->ak_done.push_back(e1); c->ak_fin.push_back(e2); }
        for (int x = 0; x < AK_NSET; ++x) if (use_fast && !c->fb_stream[x]) HIPCHK(hipStreamCreateWithFlags(&c->fb_stream[x], hipStreamNonBlocking));
        while (c->af_ev.size() < 3 * n_sub) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->af_ev.push_back(e); }
        if (use_fast) { if ((rc = c->fb_all.ensure(16 * n_sub + n_sub * (sub_reads + 1) + 16))) return rc; HIPCHK(hipMemsetAsync(c->fb_all.p, 0, 16 * n_sub * sizeof(uint32_t), c->stream)); }
        HIPCHK(hipMemsetAsync(c->ak_cursors.p, 0, (AK_CUR * n_sub + AK_CUR) * sizeof(unsigned long long), c->stream));
        HIPCHK(hipMemcpyAsync(c->ak_minscore.p, msc.data(), msc.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        t_launch[0] = mh::now_s() - t_enter;
        if ((rc = c->af_ctr_host.ensure(AF_NCTR * n_sub + AF_NCTR))) return rc;
        memset(c->af_ctr_host.p, 0, AF_NCTR * n_sub * sizeof(uint32_t));
        // host side: follows the launches
        struct SubRes { moni_aln_rec_t* recs = nullptr; uint32_t* cig = nullptr; moni_alt_t* alt = nullptr; const uint64_t* md = nullptr; const uint64_t* txt = nullptr; uint64_t nr = 0; };
        int rc_host = MONI_OK;
        auto fetch = [&](uint64_t k, SubRes& R) -> int {            // records of sub-batch k into pinned host memory
            const uint64_t r0 = sub_lo[k], nr = sub_lo[k + 1] - sub_lo[k];
            const double f0 = mh::now_s();
            HIPCHK(hipEventSynchronize(c->ak_done[k]));
            const double f1 = mh::now_s();
            if (inorder && nr) { HIPCHK(hipMemcpyAsync(c->h_recs.p + r0, c->ak_recs.p + r0, nr * sizeof(moni_aln_rec_t), hipMemcpyDeviceToHost, c->copy_stream)); HIPCHK(hipStreamSynchronize(c->copy_stream)); }
            R.recs = c->h_recs.p + r0; R.cig = c->h_cig.p + k * cig_per; R.alt = c->h_alt.p + k * alt_per; R.md = c->h_md.p + k * md_per; R.nr = nr;
            if (getenv("MONI_AK_PROFILE")) fprintf(stderr, "  sub-batch %llu: waited for the launch from %.1f to %.1f ms\n", (unsigned long long)k, (f0 - t_enter) * 1e3, (f1 - t_enter) * 1e3);
            if (force_back) for (uint64_t r = 0; r < nr; ++r) if ((r0 + r) % force_back == 0) R.recs[r].status = 2;
            if (gpu_text) {        // the text the launch wrote: copy engine transfers next to the running kernels, one per used region of the pool
                const uint64_t n_reg = use_fast ? AF_TXT_SHARDS + 1 : 1, reg_words = use_fast ? txt_per / (AF_TXT_SHARDS + 1) : txt_per;
                uint64_t used[AF_TXT_SHARDS + 1];
                for (uint64_t g = 0; g < n_reg; ++g) used[g] = 0;
                for (uint64_t r = 0; r < nr; ++r) if (R.recs[r].txt_len) {
                    const uint64_t end = R.recs[r].txt_off + ((R.recs[r].txt_len + 7) >> 3);
                    const uint64_t g = R.recs[r].txt_off / reg_words;
                    if (g >= n_reg || end > (g + 1) * reg_words) return MONI_EINVAL;
                    used[g] = std::max(used[g], end - g * reg_words);
                }
                bool any = false;
                for (uint64_t g = 0; g < n_reg; ++g) if (used[g]) {
                    HIPCHK(hipMemcpyAsync(c->h_txt.p + g * reg_words, c->ak_txt.p + k * txt_per + g * reg_words, used[g] * 8, hipMemcpyDeviceToHost, c->copy_stream));
                    any = true;
                }
                if (any) HIPCHK(hipStreamSynchronize(c->copy_stream));
                R.txt = c->h_txt.p;
            }
            { float ms = 0; if (hipEventElapsedTime(&ms, c->ak_begin[k], c->ak_done[k]) == hipSuccess) { c->dp_kernel_ms_accum += ms; c->ak_kernel_ms = ms; } }
            return MONI_OK;
        };
        // host stage of one sub-batch: MD/NM, MAPQ, SAM text; thread t writes the lines of its (contiguous) share of the reads
        auto host_stage = [&](uint64_t k, const SubRes& R) {
            const double t0 = mh::now_s();
            const uint64_t r0 = sub_lo[k], nr = R.nr;
            for (uint64_t r = 0; r < nr; ++r) if (R.recs[r].status == 2) back_of[k].push_back((uint32_t)(r0 + r));
            bool oom = false;
            mh::parallel_for(pool, nr, [&](int t, size_t lo, size_t hi) {
                mh::Aligner::OutBuf& ob = c->pieces[t];
                static_assert(sizeof(moni_alt_t) == sizeof(mh::moni_alt_like), "alt record layout");
                std::vector<char>& mds = c->md_scratch[t];
                for (size_t r = lo; r < hi; ++r) {
                    const moni_aln_rec_t& Rr = R.recs[r];
                    if (Rr.status == 2) continue;                    // handed back: filled in at the end
                    const uint64_t g = r0 + r;
                    const uint64_t off = b->offsets[g]; const uint32_t m = (uint32_t)(b->offsets[g + 1] - off);
                    if (R.txt && Rr.txt_len) {       // formatted by the kernel
                        if (!ob.ensure(Rr.txt_len)) { oom = true; break; }
                        memcpy(ob.base + ob.len, (const char*)(R.txt + Rr.txt_off), Rr.txt_len);
                        ob.len += Rr.txt_len;
                        if (Rr.status == 1) aligned_t[t]++;
                        continue;
                    }
                    if (!AL.emit_record(ob, mds, (const char*)names + name_off[g], (size_t)(name_off[g + 1] - name_off[g]), b->seq + off, quals ? quals + off : nullptr, m,
                                        Rr.status == 1, Rr.strand, Rr.ref_pos, Rr.score, Rr.score2, R.cig + Rr.cigar_off, Rr.n_cigar,
                                        (const mh::moni_alt_like*)R.alt + Rr.alt_off, Rr.n_alt,
                                        Rr.status == 1 ? (const char*)(R.md + Rr.md_off) : nullptr, Rr.md_len, Rr.nm, Rr.lift_nm)) { oom = true; break; }
                    if (Rr.status == 1) aligned_t[t]++;
                }
            });
            if (oom) eager_oom = true;
            if (eager_ok && back_of[k].empty() && !eager_oom) {
                std::vector<size_t> at(T + 1, alen);
                for (int t = 0; t < T; ++t) at[t + 1] = at[t] + c->pieces[t].len;
                const size_t need = at[T];
                if (need + 1 > acap) {
                    const size_t sub_bytes = need - alen, rest = NR - (r0 + nr);
                    size_t cap = need + (size_t)((double)sub_bytes / (double)(nr ? nr : 1) * (double)rest * 1.06) + 65536;
                    if (!grow_abuf(cap)) eager_oom = true;
                }
                if (!eager_oom) {
                    mh::parallel_for(pool, (size_t)T, [&](int, size_t lo, size_t hi) {
                        for (size_t x = lo; x < hi; ++x) { mh::Aligner::OutBuf& ob = c->pieces[x]; if (ob.len) memcpy(abuf + at[x], ob.base, ob.len); ob.len = 0; }
                    });
                    alen = need; eager_upto = k + 1;
                }
            } else {
                eager_ok = false;
                kept[k].resize(T);
                for (int t = 0; t < T; ++t) { kept[k][t].assign(c->pieces[t].base ? c->pieces[t].base : "", c->pieces[t].len); c->pieces[t].len = 0; }
            }
            host_busy += mh::now_s() - t0;
        };
        if ((int)c->md_scratch.size() < T) c->md_scratch.resize(T);
        for (int t = 0; t < T; ++t) c->pieces[t].len = 0;
        auto handle = [&](uint64_t k) {            // what the calling thread does for a finished sub-batch
            SubRes R;
            if (inorder && sub_lo[k + 1] > sub_lo[k]) {
                if (hipEventSynchronize(c->ak_done[k]) != hipSuccess) { rc_host = MONI_ENODEV; return; }
                const unsigned long long* sm = c->h_sum.p + 4 * k;
                if (sm[1] == 0 && eager_ok && !eager_oom && eager_upto == k && sm[0] <= txt_per * 8) {
                    // every line of the sub-batch was spelled by the kernels: the block goes straight into the (pinned) output buffer
                    { float ms = 0; if (hipEventElapsedTime(&ms, c->ak_begin[k], c->ak_done[k]) == hipSuccess) { c->dp_kernel_ms_accum += ms; c->ak_kernel_ms = ms; } }
                    if (k + 1 == n_sub) { t_mark[1] = mh::now_s() - t_enter; st.t_dp += mh::now_s() - t_gpu0; }
                    const double h0 = mh::now_s();
                    const size_t need = alen + (size_t)sm[0];
                    if (need + 1 > acap) {
                        const uint64_t done_reads = sub_lo[k + 1], rest = NR - done_reads;
                        const size_t cap = need + (size_t)((double)need / (double)(done_reads ? done_reads : 1) * (double)rest * 1.06) + 65536;
                        if (!grow_abuf(cap)) { eager_oom = true; return; }
                    }
                    if (sm[0]) {
                        if (hipMemcpyAsync(abuf + alen, c->ak_block.p + k * txt_per, (size_t)sm[0], hipMemcpyDeviceToHost, c->copy_stream) != hipSuccess ||
                            hipStreamSynchronize(c->copy_stream) != hipSuccess) { rc_host = MONI_ENODEV; return; }
                    }
                    alen = need; eager_upto = k + 1; aligned_t[0] += sm[2];
                    host_busy += mh::now_s() - h0;
                    return;
                }
            }
            if ((rc_host = fetch(k, R))) return;
            if (k + 1 == n_sub) { t_mark[1] = mh::now_s() - t_enter; st.t_dp += mh::now_s() - t_gpu0; }
            host_stage(k, R);
        };
        uint64_t retired = 0, launched = 0;
        auto retire = [&](bool block) {            // finished sub-batches in order; block = false: only those already through
            while (retired < launched && !rc_host && !eager_oom) {
                if (!block && sub_lo[retired + 1] > sub_lo[retired] && hipEventQuery(c->ak_done[retired]) != hipSuccess) break;
                handle(retired);
                ++retired;
            }
        };
        for (uint64_t k = 0; k < n_sub; ++k) {
            const uint64_t r0 = sub_lo[k], nr = sub_lo[k + 1] - sub_lo[k];
            uint64_t n_waves = ak_waves;
            if (!use_fast && n_waves * AK_NL > nr) n_waves = (nr + AK_NL - 1) / AK_NL;
            waves_used = std::max(waves_used, n_waves);
            ak_args_t A;
            memset(&A, 0, sizeof A);
            A.P.min_len = prm->min_len; A.P.ext_len = prm->ext_len; A.P.check_k = prm->check_k; A.P.region_dist = prm->region_dist;
            A.P.filter_freq = prm->filter_freq; A.P.left_mem_check = prm->left_mem_check; A.P.freq_thr = prm->freq_thr;
            A.P.smatch = prm->smatch; A.P.gapo = prm->gapo; A.P.gapo2 = prm->gapo2; A.P.gape = prm->gape; A.P.gape2 = prm->gape2;
            A.P.max_dist_x = prm->max_dist_x; A.P.max_dist_y = prm->max_dist_y; A.P.max_iter = prm->max_iter; A.P.max_pred = prm->max_pred;
            A.P.min_chain_score = prm->min_chain_score; A.P.min_chain_length = prm->min_chain_length;
            A.P.n_text = I->K.n_text; A.P.n_seq = I->K.n_seq; A.P.seq_starts = I->d_seq_starts;
            A.P.lift_seqs = I->d_lift_seqs; A.P.lift_runs = I->d_lift_runs; A.P.pdir = I->d_pdir;
            A.D.sc_mch = prm->smatch; A.D.sc_mis = -prm->smismatch; A.D.sc_N = -prm->gape; A.D.wild = 4; A.D.qo = prm->gapo; A.D.e = prm->gape;
            A.D.end_bonus = prm->end_bonus; A.D.reads = c->seq.p; A.D.text = I->d_text; A.D.n_text = I->K.n_text;
            A.D.reads_limit = (c->total_len + 8) & ~7ull; A.D.text_limit = (I->K.n_text + 8) & ~7ull;       // both buffers carry 16 bytes of padding
            A.mems = c->mems.p; A.occs = c->occs.p; A.read_mem_off = c->read_mem_off.p; A.offs = c->offs.p;
            A.min_score_of_len = c->ak_minscore.p; A.max_len = (uint32_t)c->max_len + 1; A.read_lo = r0; A.n_reads = nr;
            A.slots = c->ak_slots.p + (k % AK_NSET) * ak_waves * AK_NL; A.waves = c->ak_waves.p + (k % AK_NSET) * ak_waves;
            // records: pinned host memory when host threads read them behind every launch; with the lines ordered on the GPU they are read only
            // when a sub-batch has hand-backs, and stay in HBM until then (80 bytes per read over PCIe were what finish_wave_kernel waited for)
            A.recs = (inorder ? c->ak_recs.p : c->h_recs.p) + r0; A.cig_pool = c->h_cig.p + k * cig_per; A.cig_cap = cig_per; A.alt_pool = c->h_alt.p + k * alt_per;
            A.alt_cap = alt_per; A.md_pool = c->h_md.p + k * md_per; A.md_cap = md_per; A.cursors = c->ak_cursors.p + AK_CUR * k;
            if (gpu_text) {
                A.fmt.rnames = c->ak_rnames.p; A.fmt.rname_off = c->ak_rname_off.p; A.fmt.quals = quals ? c->ak_quals.p : nullptr;
                A.fmt.snames = I->d_snames; A.fmt.sname_off = I->d_sname_off; A.fmt.mapq_tab = c->ak_mapq_tab.p; A.fmt.mapq_tab_n = 8192;
                A.fmt.min_len = (int32_t)prm->min_len; A.fmt.smatch = prm->smatch; A.fmt.smismatch = prm->smismatch;
                A.fmt.txt_pool = c->ak_txt.p + k * txt_per; A.fmt.txt_cap = (use_fast && nr > 0) ? txt_per / (AF_TXT_SHARDS + 1) : txt_per;
            }
            if (inorder) { A.dev_len = c->ak_dev_len.p + r0 + k; A.dev_off = c->ak_dev_off.p + r0 + k; A.dev_sum = c->ak_dev_sum.p + 160 * k; }
            hipStream_t sx = c->ak_stream[k % AK_NSET];
            // (the set's buffers are free once its previous sub-batch is through finish_wave_kernel - stream order; align_kernel and the gather of that
            // sub-batch read only per-sub-batch buffers and may still be running on the hand-over stream: a slow hand-over read delays its own
            // sub-batch's block, not the launches behind it)
            HIPCHK(hipEventRecord(c->ak_begin[k], sx));
            bool done_recorded = false;
            if (use_fast && nr > 0) {
                moni_ctx::AfSet& S = c->af[k % AK_NSET];
                af_args_t G;
                memset(&G, 0, sizeof G);
                G.A = A;
                G.plans = S.plans.p; G.tasks = S.tasks.p; G.task_cap = af_slot_cap; G.ntasks = S.ntasks.p; G.res = S.res.p; G.bin_q = S.bin_q.p; G.bin_cap = af_task_cap; G.task_pos = S.task_pos.p;
                G.chunks = S.chunks.p; G.chunk_cap = af_chunk_cap; G.dirs = S.dirs.p; G.dirs_cap = af_dirs_cap; G.tb_task = S.tb_task.p; G.tb = S.tb.p; G.tb_cap = af_tb_cap;
                G.fb_n = c->fb_all.p + 16 * k; G.fb_list = c->fb_all.p + 16 * n_sub + k * (sub_reads + 1); G.big_list = S.big_list.p; G.huge_list = S.big_list.p + (sub_reads + 1); G.list0 = S.big_list.p + 2 * (sub_reads + 1); G.fin_scratch = S.fin.p; G.fin_stride = sizeof(af_fin_t); G.ctr = S.ctr.p;
                G.bnd = S.bnd.p;
                G.pat = c->pat.p; G.blk = c->blk.p; G.text2 = I->d_text2; G.exc = I->d_exc; G.exc_sh = I->exc_sh; G.pflag = c->pflag.p; G.wave_max = af_wave_max();
                G.txt_cur = S.txt_cur.p; G.txt_shard_words = txt_per / (AF_TXT_SHARDS + 1);
                HIPCHK(hipMemsetAsync(S.txt_cur.p, 0, AF_TXT_SHARDS * 8 * sizeof(unsigned long long), sx));
#ifdef AF_PROFILE
                if ((rc = S.prof.ensure(32))) return rc;
                if (k < AK_NSET) HIPCHK(hipMemsetAsync(S.prof.p, 0, 32 * 8, sx));
                G.prof = S.prof.p;
                if (const char* v = getenv("MONI_AF_DBG")) G.dbg = (uint32_t)atoi(v);
#endif
#ifdef AF_CUTS
                if (const char* v = getenv("MONI_AF_DBG")) G.dbg = (uint32_t)atoi(v);
#endif
                if (const char* v = getenv("MONI_AF_DBG")) G.dbg |= (uint32_t)atoi(v) & (64u | 65536u | 131072u);          // any build: 64 the serial anchor sort (cross-check), 65536 every global problem through the full-matrix kernel, 131072 every extension / gap fill through the tile kernels
                HIPCHK(hipMemsetAsync(S.ctr.p, 0, AF_NCTR * sizeof(uint32_t), sx));
                // LEVEL 0's instance: the small one, or - reads of more than 200 bases, whose seeds have more occurrences than it holds - the middle one
                static const int l0_force = getenv("MONI_AF_L0") ? atoi(getenv("MONI_AF_L0")) : -1;          // 0 small, 1 middle
                const bool l0_mid = l0_force >= 0 ? l0_force == 1 : c->max_len > 200;
                G.l0_mm = l0_mid ? (uint32_t)af_wave_mid_t::MM : (uint32_t)af_wave_small_t::MM; G.l0_ma = l0_mid ? (uint32_t)af_wave_mid_t::MA : (uint32_t)af_wave_small_t::MA;
                const bool no_k2 = getenv("MONI_AF_NOPLANK") != nullptr;          // (measurement: the LEVEL-0 instance runs the selection loop and builds the plan itself, as the others do)
                G.ctab = (no_k2 || l0_mid) ? nullptr : S.ctab.p;
                hipLaunchKernelGGL(classify_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, sx, G);
                if (l0_mid) {
                    const dim3 g1((unsigned)std::min<uint64_t>(nr, (uint64_t)n_cu * 4 * 4));
                    hipLaunchKernelGGL((chain_plan_kernel<af_wave_mid_t, 0, 4>), g1, dim3(64), 0, sx, G);
                } else {
                    // four reads per wavefront (16 lanes each, 4 x 4.6 KB of LDS): 8 wavefronts = 32 reads in flight per CU, as with one read per wavefront at
                    // 8 waves/SIMD, but the one-lane sections issue a quarter of the instructions.  MONI_AF_GW=64: one read per wavefront (MONI_AF_K1OCC waves/SIMD)
                    static const int l0_gw = getenv("MONI_AF_GW") ? atoi(getenv("MONI_AF_GW")) : 64;
                    static const int k1occ = getenv("MONI_AF_K1OCC") ? atoi(getenv("MONI_AF_K1OCC")) : (l0_gw == 16 ? 2 : 8);      // one read per wavefront: measured 5, 6, 8 waves/SIMD
                    const dim3 g1((unsigned)std::min<uint64_t>(l0_gw == 16 ? (nr + 3) / 4 : nr, (uint64_t)n_cu * 4 * k1occ));
                    if (l0_gw == 16) hipLaunchKernelGGL((chain_plan_kernel<af_wave_small_t, 0, 2, 16>), g1, dim3(64), 0, sx, G);
                    else if (k1occ == 4) hipLaunchKernelGGL((chain_plan_kernel<af_wave_small_t, 0, 4>), g1, dim3(64), 0, sx, G);
                    else if (k1occ == 5) hipLaunchKernelGGL((chain_plan_kernel<af_wave_small_t, 0, 5>), g1, dim3(64), 0, sx, G);
                    else if (k1occ == 8) hipLaunchKernelGGL((chain_plan_kernel<af_wave_small_t, 0, 8>), g1, dim3(64), 0, sx, G);
                    else hipLaunchKernelGGL((chain_plan_kernel<af_wave_small_t, 0, 6>), g1, dim3(64), 0, sx, G);
                }
                hipLaunchKernelGGL((chain_plan_kernel<af_wave_t, 1>), dim3((unsigned)std::min<uint64_t>(nr, (uint64_t)n_cu * 10)), dim3(64), 0, sx, G);
                static const bool no_huge = getenv("MONI_AF_NOHUGE") != nullptr;          // (debugging aid: reads beyond the large instance then go straight to align_kernel)
                if (!no_huge) hipLaunchKernelGGL((chain_plan_kernel<af_wave_huge_t, 2>), dim3((unsigned)std::min<uint64_t>(nr, (uint64_t)n_cu)), dim3(64), 0, sx, G);      // ~90 KB of LDS per wave: one per CU
                HIPCHK(hipGetLastError());
                if (G.ctab) hipLaunchKernelGGL(plan_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, sx, G);
                hipLaunchKernelGGL(bin_tasks_kernel, dim3((unsigned)((nr + AF_BT_READS - 1) / AF_BT_READS)), dim3(256), 0, sx, G);
                HIPCHK(hipEventRecord(c->af_ev[3 * k], sx));
                af_launch_dp(G, sx, af_dp_grid, (unsigned)n_cu, nr);
                HIPCHK(hipEventRecord(c->af_ev[3 * k + 1], sx));
                hipLaunchKernelGGL(select_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, sx, G);
                hipLaunchKernelGGL(traceback_kernel, dim3((unsigned)((af_tb_cap + 255) / 256)), dim3(256), 0, sx, G);
                HIPCHK(hipEventRecord(c->af_ev[3 * k + 2], sx));
                // the reads the staged kernels handed over: few, but each a long serial job (milliseconds): align_kernel takes them on its own stream
                // beside finish_wave_kernel, which hands nothing over any more (a line or CIGAR beyond its staging goes to the host pipeline)
                A.read_list = G.fb_list; A.n_reads_dev = G.fb_n;
                hipStream_t sf = c->fb_stream[k % AK_NSET];
                HIPCHK(hipStreamWaitEvent(sf, c->af_ev[3 * k + 2], 0));
                hipLaunchKernelGGL(align_kernel, dim3((unsigned)ak_waves), dim3(64), 0, sf, A);
                // the record and the SAM line of every read that stayed on the staged path: one wave per read when the kernel spells the text
                static const int fin_mult = getenv("MONI_AF_FINGRID") ? atoi(getenv("MONI_AF_FINGRID")) : 96;          // blocks per CU: 4x what is resident (24 by LDS), so that the strided share of a block is short and the tail even (measured 24 .. 768)
                if (gpu_text && !fin_v1) {          // a lane per read for the serial work (CIGAR, lift, MD / NM, MAPQ -> a recipe), then a wavefront per read for the line
                    hipLaunchKernelGGL(finish_prep_kernel, dim3((unsigned)std::min<uint64_t>((nr + 63) / 64, (uint64_t)n_cu * 32)), dim3(64), 0, sx, G, S.recipes.p);
                    hipLaunchKernelGGL(finish_render_kernel, dim3((unsigned)std::min<uint64_t>(nr, (uint64_t)n_cu * fin_mult)), dim3(64), 0, sx, G, (const uint32_t*)S.recipes.p);
                } else if (gpu_text) hipLaunchKernelGGL(finish_wave_kernel, dim3((unsigned)std::min<uint64_t>(nr, (uint64_t)n_cu * fin_mult)), dim3(64), 0, sx, G);
                else hipLaunchKernelGGL(finish_kernel, dim3((unsigned)std::min<uint64_t>((nr + 63) / 64, af_fin_grid)), dim3(64), 0, sx, G);
                HIPCHK(hipMemcpyAsync(c->af_ctr_host.p + AF_NCTR * k, S.ctr.p, AF_NCTR * sizeof(uint32_t), hipMemcpyDeviceToHost, sx));      // (before the set's next sub-batch clears them)
                HIPCHK(hipEventRecord(c->ak_fin[k], sx));
                HIPCHK(hipStreamWaitEvent(sf, c->ak_fin[k], 0));
                if (inorder) {          // lines in read order: scan of the lengths, gather, summary for the host
                    uint64_t* pos = c->ak_dev_pos.p + r0 + 2 * k;
                    size_t tmp_bytes = 0;
                    if (rocprim::exclusive_scan(nullptr, tmp_bytes, A.dev_len, pos, (uint64_t)0, nr, rocprim::plus<uint64_t>(), sf) != hipSuccess || tmp_bytes > gather_tmp_bytes + 16) return MONI_ENODEV;
                    if (rocprim::exclusive_scan(c->gather_tmp[k % AK_NSET].p, tmp_bytes, A.dev_len, pos, (uint64_t)0, nr, rocprim::plus<uint64_t>(), sf) != hipSuccess) return MONI_ENODEV;
                    hipLaunchKernelGGL(gather_lines_kernel, dim3((unsigned)std::min<uint64_t>((nr + 3) / 4, (uint64_t)n_cu * 8)), dim3(256), 0, sf, (const uint64_t*)A.fmt.txt_pool,
                                       (const uint64_t*)A.dev_len, (const uint64_t*)A.dev_off, (const uint64_t*)pos, nr, reinterpret_cast<uint8_t*>(c->ak_block.p + k * txt_per));
                    hipLaunchKernelGGL(gather_summary_kernel, dim3(1), dim3(64), 0, sf, (const uint64_t*)A.dev_len, (const uint64_t*)pos, nr, (const unsigned long long*)A.dev_sum,
                                       c->ak_dev_sum.p + 160 * k + 150);
                    HIPCHK(hipMemcpyAsync(c->h_sum.p + 4 * k, c->ak_dev_sum.p + 160 * k + 150, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, sf));
                }
                HIPCHK(hipEventRecord(c->ak_done[k], sf));
                done_recorded = true;
            } else if (nr > 0) {
                hipLaunchKernelGGL(align_kernel, dim3((unsigned)n_waves), dim3(64), 0, sx, A);
            }
            if (!done_recorded) HIPCHK(hipEventRecord(c->ak_done[k], sx));
            HIPCHK(hipGetLastError());
            launched = k + 1;
        }
        t_launch[1] = mh::now_s() - t_enter;
        if (rc_host == MONI_OK) retire(true);
        if (rc_host) { for (int x = 0; x < AK_NSET; ++x) { (void)hipStreamSynchronize(c->ak_stream[x]); if (c->fb_stream[x]) (void)hipStreamSynchronize(c->fb_stream[x]); } drop_abuf(); return rc_host; }
        t_mark[2] = mh::now_s() - t_enter;
        st.dp_rounds = n_sub;          // align_kernel launches
        bool any_dirs_ovf_batch = false;
        if (n_sub) {        // statistics of all launches, once the GPU is idle
            std::vector<unsigned long long> cur(AK_CUR * n_sub);
            HIPCHK(hipMemcpy(cur.data(), c->ak_cursors.p, cur.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::vector<uint32_t> fbn(16 * n_sub, 0);
            if (use_fast) HIPCHK(hipMemcpy(fbn.data(), c->fb_all.p, fbn.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            for (uint64_t k = 0; k < n_sub; ++k) {
                const unsigned long long* q = cur.data() + AK_CUR * k;
                st.dp_tasks += q[2]; st.dp_cells += q[3]; st.dp_reused += q[8]; st.dp_cells_reused += q[9];
                if (use_fast && sub_lo[k + 1] > sub_lo[k]) {         // HIP-event times of the staged kernels' groups (they overlap with the other stream's: sums, not spans)
                    float ms = 0;
                    if (hipEventElapsedTime(&ms, c->ak_begin[k], c->af_ev[3 * k]) == hipSuccess) st.t_k_chain += ms / 1e3;
                    if (hipEventElapsedTime(&ms, c->af_ev[3 * k], c->af_ev[3 * k + 1]) == hipSuccess) st.t_k_dp += ms / 1e3;
                    if (hipEventElapsedTime(&ms, c->af_ev[3 * k + 1], c->af_ev[3 * k + 2]) == hipSuccess) st.t_k_select += ms / 1e3;
                    if (hipEventElapsedTime(&ms, c->af_ev[3 * k + 2], c->ak_fin[k]) == hipSuccess) st.t_k_finish += ms / 1e3;
                }
                if (use_fast) {         // the staged kernels' own counters
                    const uint32_t* fc = c->af_ctr_host.p + AF_NCTR * k;
                    unsigned long long cells, rb, cutc, slots; memcpy(&cells, fc + AFC_CELLS, 8); memcpy(&rb, fc + AFC_RBYTES, 8); memcpy(&cutc, fc + AFC_CUTCELLS, 8); memcpy(&slots, fc + AFC_SLOTS, 8);
                    st.dp_cells_cut += cutc; st.dp_slots += slots;
                    any_dirs_ovf_batch = any_dirs_ovf_batch || fc[AFC_DIRS_OVF] != 0;
                    st.dp_tasks += fc[AFC_NT] + (fc[AFC_TASKS] >= (uint32_t)(sub_lo[k + 1] - sub_lo[k]) * AF_MAX_TASKS_READ ? fc[AFC_TASKS] - (uint32_t)(sub_lo[k + 1] - sub_lo[k]) * AF_MAX_TASKS_READ : 0u); st.dp_cells += cells; st.kernel_fallback += fbn[16 * k]; st.dp_ref_bytes += rb;
                    for (int x = 0; x < AF_WHY_N; ++x) why_sum[x] += fc[AFC_WHY + x];
                    if (getenv("MONI_AK_PROFILE")) { unsigned long long du; memcpy(&du, fc + AFC_DIROFF, 8); fprintf(stderr, "  direction bits of sub-batch %llu: %.1f MB used of %.1f MB\n", (unsigned long long)k, du / 1e6, af_dirs_cap / 1e6); }
                    if (getenv("MONI_AK_PROFILE")) fprintf(stderr, "  staged kernels, sub-batch %llu: %u DP tasks, %llu cells, %u traced, %u large + %u small + %u global chunks, %u reads to align_kernel%s\n",
                                                           (unsigned long long)k, fc[AFC_NT], cells, fc[AFC_TRACED], fc[AFC_NCHUNKS], fc[AFC_NCHUNKS + 1], fc[AFC_NCHUNKS + 2], fbn[16 * k], fc[AFC_DIRS_OVF] ? " (direction bytes overflowed)" : "");
                    if (getenv("MONI_AK_PROFILE")) { fprintf(stderr, "    DP problems per bin (16 of the large tile by query length <= 8 13 16 24 32 48 64 .. 160 192 224 256, 3 of the small tile <= 8 16 32, 16 global by query length / 16):"); for (int x = 0; x < AF_NBIN; ++x) fprintf(stderr, " %u", fc[AFC_BINS + x]); fprintf(stderr, "\n"); }
#if defined(AF_CUTS)
                    if (getenv("MONI_AK_PROFILE")) { fprintf(stderr, "    tile problems (extension | gap fill q = t | gap fill q != t) x (band <= 16 | wider | extension with t < q | no bound):"); for (int x = 0; x < 12; ++x) fprintf(stderr, "%s %u", x % 4 == 0 ? " |" : "", fc[180 + x]); fprintf(stderr, "\n"); }
#endif
                    if (getenv("MONI_AK_PROFILE")) { fprintf(stderr, "    global problems by band width / 4 (0-3, 4-7, ... , >= 52):"); for (int x = 0; x < 14; ++x) fprintf(stderr, " %u", fc[AFC_BANDH + x]); fprintf(stderr, "\n"); }
                    if (getenv("MONI_AK_PROFILE")) fprintf(stderr, "    handed over because: long read %u, anchors/seeds %u, chains %u, chains to score %u, chain length %u, DP size %u, overlapping anchors %u, wildcard %u, "
                                                           "loop depends on a score %u, extension short of the query end %u, capacity %u, CIGAR %u\n",
                                                           fc[AFC_WHY + 0], fc[AFC_WHY + 1], fc[AFC_WHY + 2], fc[AFC_WHY + 3], fc[AFC_WHY + 4], fc[AFC_WHY + 5], fc[AFC_WHY + 6], fc[AFC_WHY + 7], fc[AFC_WHY + 8], fc[AFC_WHY + 9], fc[AFC_WHY + 10], fc[AFC_WHY + 11]);
                }
                prof[0] += (double)q[5]; prof[1] += (double)q[6]; prof[2] += (double)q[7]; for (int x = 0; x < 4; ++x) prof[3 + x] += (double)q[10 + x];
                for (int x = 0; x < 8; ++x) hist[x] += (double)q[16 + x];
                for (int x = 0; x < 7; ++x) cyc[x] += (double)q[24 + x];
            }
        }
        if (any_dirs_ovf_batch && c->dirs_scale < 32) c->dirs_scale *= 2;
        static_assert(AF_WHY_N == 12, "moni_align_stats_t::handover_why");
        for (int x = 0; x < AF_WHY_N; ++x) why_out[x] = why_sum[x];
#ifdef AF_PROFILE
        if (use_fast && n_sub) { unsigned long long pf[32]; for (int x = 0; x < (int)std::min<uint64_t>(n_sub, AK_NSET); ++x) { HIPCHK(hipMemcpy(pf, c->af[x].prof.p, sizeof pf, hipMemcpyDeviceToHost));
            fprintf(stderr, "chain_plan_kernel wave cycles (set %d): load+filter+anchors %.3g, chain %.3g (sort %.3g, dp %.3g, ends+backtrack %.3g), lifts %.3g, plan %.3g, whole read %.3g\n", x,
                    (double)pf[0], (double)pf[1], (double)pf[5], (double)pf[6], (double)pf[7], (double)pf[2], (double)pf[3], (double)pf[4]);
            fprintf(stderr, "finish_wave_kernel wave cycles (set %d): stage + stitch + lift %.3g, MD / NM / MAPQ %.3g, segment list %.3g, render %.3g, out %.3g, whole read %.3g\n", x,
                    (double)pf[16], (double)pf[17], (double)pf[18], (double)pf[19], (double)pf[21], (double)pf[22]); } }
#endif
        double ak_sum_ms = c->dp_kernel_ms_accum;
        if (n_sub) { float ms = 0; if (hipEventElapsedTime(&ms, c->ak_begin[0], c->ak_done[n_sub - 1]) == hipSuccess) c->dp_kernel_ms_accum = ms; }      // launches overlap: report the span
        if (getenv("MONI_AK_PROFILE")) fprintf(stderr, "align_kernel: %.3f ms span (%.3f ms summed) in %llu launches, %llu waves x %d reads in flight; wave cycles: take+chain %.3g, later drives %.3g, DP %.3g; lane cycles in ac_init: load %.3g sort %.3g chain-dp %.3g backtrack %.3g\n", c->dp_kernel_ms_accum, ak_sum_ms, (unsigned long long)n_sub, (unsigned long long)waves_used, (int)AK_NL, prof[0], prof[1], prof[2], prof[3], prof[4], prof[5], prof[6]);
        if (getenv("MONI_AK_PROFILE")) fprintf(stderr, "DP problems run, by live rows min(qlen,tlen) <=16 / <=32 / <=64 / >64: count %.3g %.3g %.3g %.3g, cells %.3g %.3g %.3g %.3g\n",
                                               hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
        if (getenv("MONI_AK_PROFILE")) fprintf(stderr, "wave cycles inside the DP phase: per-read setup %.3g, memo lookups %.3g, DP by live rows <=16 / <=32 / <=64 / >64: %.3g %.3g %.3g %.3g; 1x1 problems answered in closed form: %.3g\n",
                                               cyc[0], cyc[1], cyc[2], cyc[3], cyc[4], cyc[5], cyc[6]);
        st.dp_reused += (uint64_t)cyc[6];
        double t0 = mh::now_s();
        for (int t = 0; t < T; ++t) st.aligned += aligned_t[t];
        st.reads = NR;
        std::vector<uint32_t> back;
        for (auto& v : back_of) back.insert(back.end(), v.begin(), v.end());
        std::vector<std::string> back_line(back.size());
        if (!back.empty()) {
            // the reads the kernel handed back, as one batch through the host pipeline
            std::vector<uint8_t> sseq, snames, squal; std::vector<uint64_t> soff(1, 0), snoff(1, 0);
            for (uint32_t r : back) {
                const uint64_t off = b->offsets[r], m = b->offsets[r + 1] - b->offsets[r];
                sseq.insert(sseq.end(), b->seq + off, b->seq + off + m); soff.push_back(sseq.size());
                snames.insert(snames.end(), names + name_off[r], names + name_off[r + 1]); snoff.push_back(snames.size());
                if (quals) squal.insert(squal.end(), quals + off, quals + off + m);
            }
            moni_read_batch_t sub{sseq.data(), soff.data(), (uint64_t)back.size()};
            std::string sout; mh::AlignStats s2;
            if ((rc = host_align_subset(c, *prm, sub, snames.data(), snoff.data(), quals ? squal.data() : nullptr, sout, s2))) { drop_abuf(); return rc; }
            size_t p0 = 0;
            for (size_t x = 0; x < back.size(); ++x) { const size_t p1 = sout.find('\n', p0); back_line[x] = sout.substr(p0, p1 - p0 + 1); p0 = p1 + 1; }
            st.aligned += s2.aligned; st.dp_tasks += s2.dp_tasks; st.dp_cells += s2.dp_cells; st.dp_rounds += s2.dp_rounds;
            st.t_chain += s2.t_chain;     // counts the whole fallback as "chain/host" time below
        }
        st.handed_back = back.size();
        if (eager_oom) { drop_abuf(); return MONI_ENOMEM; }
        if (back.empty() && eager_upto == n_sub && (abuf || NR == 0)) {
            if (!abuf && !grow_abuf(64)) return MONI_ENOMEM;
            abuf[alen] = 0;
            *sam = abuf; *sam_len = alen;
            out_done = true;
        } else {
            // rare: splice the handed-back lines in at their read positions (a thread's piece is split at those reads)
            if (abuf) { out.assign(abuf, alen); drop_abuf(); }
            size_t bi = 0;
            for (uint64_t k = eager_upto; k < n_sub; ++k) {
                const uint64_t r0 = sub_lo[k], nr = sub_lo[k + 1] - sub_lo[k];
                for (int t = 0; t < T; ++t) {
                    const size_t lo = (nr < 2 || T <= 1) ? (t == 0 ? 0 : nr) : nr * t / T, hi = (nr < 2 || T <= 1) ? (t == 0 ? nr : nr) : nr * (t + 1) / T;
                    const std::string& piece = kept[k][t];
                    size_t p0 = 0;
                    for (size_t r = lo; r < hi; ++r) {
                        if (bi < back.size() && back[bi] == r0 + r) { out += back_line[bi++]; continue; }
                        const size_t p1 = piece.find('\n', p0);
                        out.append(piece, p0, p1 - p0 + 1); p0 = p1 + 1;
                    }
                }
            }
        }
        st.t_host += host_busy + (mh::now_s() - t0);
        if (getenv("MONI_AK_PROFILE")) fprintf(stderr, "align_core wall: launches queued from %.1f to %.1f ms\n", t_launch[0] * 1e3, t_launch[1] * 1e3);
        if (getenv("MONI_AK_PROFILE")) fprintf(stderr, "align_core wall: seeded at %.1f ms, last launch done at %.1f, last host stage done at %.1f, text ready at %.1f\n",
                                               t_mark[0] * 1e3, t_mark[1] * 1e3, t_mark[2] * 1e3, (mh::now_s() - t_enter) * 1e3);
    }
    if (!out_done) {
        char* dst;
        if (ctx_out) {
            if (c->out_cap < out.size() + 1) {
                char* nb = nullptr;
                if (hipHostMalloc((void**)&nb, out.size() + 1, hipHostMallocDefault) != hipSuccess) return MONI_ENOMEM;
                if (c->out_buf) (void)hipHostFree(c->out_buf);
                c->out_buf = nb; c->out_cap = out.size() + 1;
            }
            dst = c->out_buf;
        } else if (!(dst = (char*)malloc(out.size() + 1))) return MONI_ENOMEM;
        memcpy(dst, out.data(), out.size());
        dst[out.size()] = 0;
        *sam = dst; *sam_len = out.size();
    }
    if (stats) {
        stats->reads = st.reads; stats->aligned = st.aligned; stats->dp_tasks = st.dp_tasks; stats->dp_cells = st.dp_cells; stats->dp_rounds = st.dp_rounds;
        stats->t_seed = st.t_seed; stats->t_chain = st.t_chain; stats->t_dp = st.t_dp; stats->t_host = st.t_host;
        stats->t_dp_kernel = c->dp_kernel_ms_accum / 1e3;
        stats->handed_back = st.handed_back; stats->dp_reused = st.dp_reused; stats->dp_cells_reused = st.dp_cells_reused;
        stats->kernel_fallback = st.kernel_fallback; stats->dp_ref_bytes = st.dp_ref_bytes; stats->dp_cells_cut = st.dp_cells_cut; stats->dp_slots = st.dp_slots;
        stats->t_k_chain = st.t_k_chain; stats->t_k_dp = st.t_k_dp; stats->t_k_select = st.t_k_select; stats->t_k_finish = st.t_k_finish;
        for (int x = 0; x < 12; ++x) stats->handover_why[x] = why_out[x];
    }
    return MONI_OK;
}

// aligner::align with report_mems (aligner_ksw2.hpp:346-373): one secondary SAM record per occurrence of every MEM (no halves), the
// record's SEQ / QUAL being the MEM's part of the read (of its reverse complement for the other strand)
int moni_report_mems_batch(moni_ctx_t* c, const moni_read_batch_t* b, const uint8_t* names, const uint64_t* name_off, const uint8_t* quals,
                           const moni_align_params_t* prm, char** sam, uint64_t* sam_len) {
    if (!c || !b || !prm || !sam || !sam_len || (b->n_reads && (!names || !name_off))) return MONI_EINVAL;
    int rc = moni_reads_upload(c, b);
    if (rc) return rc;
    moni_seed_params_t sp;
    sp.min_len = prm->min_len; sp.filter_seeds = prm->filter_seeds; sp.n_seeds_thr = prm->n_seeds_thr; sp.report_mems = 1;
    if ((rc = moni_seed_run(c, &sp))) return rc;
    try {
        std::vector<moni_mem_t> mems(c->n_mems); std::vector<uint64_t> occs(c->n_occs), rmo(c->n_reads + 1);
        if ((rc = moni_seed_fetch(c, mems.data(), occs.data(), rmo.data()))) return rc;
        const mh::HostIndex& ix = c->idx->hix;
        std::string out;
        char num[32];
        for (uint64_t r = 0; r < c->n_reads; ++r) {
            const uint64_t off = b->offsets[r], m = b->offsets[r + 1] - off;
            size_t total = 0;
            for (uint64_t k = rmo[r]; k < rmo[r + 1]; ++k) total += mems[k].occ_cnt;
            for (uint64_t k = rmo[r]; k < rmo[r + 1]; ++k) {
                const moni_mem_t& M = mems[k];
                if (prm->filter_freq) { const double fr = static_cast<double>(M.occ_cnt) / total; if (fr > prm->freq_thr) continue; }      // seed_freq_filter
                const bool rc_strand = (M.mate & 2) != 0;
                std::string seq(M.len, 'N'), ql;
                for (uint32_t x = 0; x < M.len; ++x) seq[x] = rc_strand ? (char)mh::compl_of(b->seq[off + m - 1 - (M.idx + x)]) : (char)b->seq[off + M.idx + x];
                if (quals) { ql.resize(M.len); for (uint32_t x = 0; x < M.len; ++x) ql[x] = rc_strand ? (char)quals[off + m - 1 - (M.idx + x)] : (char)quals[off + M.idx + x]; }
                for (uint32_t j = 0; j < M.occ_cnt; ++j) {
                    const auto ref = ix.index(occs[M.occ_off + j]);
                    out.append((const char*)names + name_off[r], (size_t)(name_off[r + 1] - name_off[r]));
                    out += rc_strand ? "\t272\t" : "\t256\t";
                    out += ix.names[ref.first]; out.push_back('\t');
                    snprintf(num, sizeof num, "%d", (int)(ref.second + 1)); out += num;
                    out += "\t255\t"; snprintf(num, sizeof num, "%d", (int)M.len); out += num; out += "M\t*\t0\t0\t";
                    out += seq; out.push_back('\t');
                    if (quals) out += ql; else out.push_back('*');
                    out += "\tAS:i:0\tNM:i:0\tMD:Z:\tOA:Z:*,0,"; out += rc_strand ? "-" : "+"; out += ",*,255,0;\tAA:Z:\n";
                }
            }
        }
        char* dst = (char*)malloc(out.size() + 1);
        if (!dst) return MONI_ENOMEM;
        memcpy(dst, out.data(), out.size()); dst[out.size()] = 0;
        *sam = dst; *sam_len = out.size();
        return MONI_OK;
    } catch (const std::bad_alloc&) { return MONI_ENOMEM; }
}

// ---- the reference's .ldx (liftidx) --------------------------------------------------------------------------------------------
__global__ void lift_batch_kernel(const ac_params_t P, const uint64_t* __restrict__ pos, uint64_t n, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = ac_lift(P, pos[i]);
}

// ---- <prefix>.thrbv.full.lcp.ms (ms_index_io.hpp) --------------------------------------------------------------------------------------
int moni_ms_file_info(const char* path, uint64_t* n, uint64_t* r) {
    if (!path) return MONI_EINVAL;
    try {          // (file-facing entry points: an allocation failure must not cross the C ABI as an exception)
        uint64_t a = 0, b = 0;
        int rc = msio::info_ms(path, a, b);
        if (rc) return rc;
        if (n) *n = a;
        if (r) *r = b;
        return MONI_OK;
    } catch (const std::bad_alloc&) { return MONI_ENOMEM; }
}

int moni_ms_file_read(const char* path, uint64_t r, uint64_t* F, uint8_t* heads, uint64_t* starts, uint64_t* ssa, uint64_t* esa, uint64_t* thr, uint64_t* slcp,
                      char* err, uint64_t err_cap) {
    if (!path || !F || !heads || !starts || !ssa || !esa || !thr || !slcp) return MONI_EINVAL;
    try {
        msio::MsFile M; std::string e;
        int rc = msio::load_ms(path, M, e);
        if (rc) { if (err && err_cap) snprintf(err, (size_t)err_cap, "%s", e.c_str()); return rc; }
        if (M.r != r) return MONI_EINVAL;
        memcpy(F, M.F.data(), 256 * 8); memcpy(heads, M.heads.data(), r); memcpy(starts, M.starts.data(), (r + 1) * 8);
        memcpy(ssa, M.ssa.data(), r * 8); memcpy(esa, M.esa.data(), r * 8); memcpy(thr, M.thr.data(), r * 8); memcpy(slcp, M.slcp.data(), r * 8);
        return MONI_OK;
    } catch (const std::bad_alloc&) { return MONI_ENOMEM; }
}

int moni_ms_file_write(const moni_flat_index_t* f, const char* path) {
    if (!f || !path || !f->F || !f->heads || !f->starts || !f->ssa || !f->esa || !f->thr || f->r < 1 || f->n < 2) return MONI_EINVAL;      // slcp NULL: the .thrbv.full.ms form
    try { return msio::save_ms(path, *f, f->r); } catch (const std::bad_alloc&) { return MONI_ENOMEM; }
}

int moni_index_load_reference(const char* ms_path, const char* ldx_path, const char* text_path, int device, moni_index_t** out) {
    if (!ms_path || !ldx_path || !out) return MONI_EINVAL;
    try {
        msio::MsFile M; std::string e;
        int rc = msio::load_ms(ms_path, M, e);
        if (rc) { fprintf(stderr, "moni_hip: %s: %s\n", ms_path, e.c_str()); return rc; }
        refio::Ldx L;
        if ((rc = refio::load_ldx(ldx_path, L))) { fprintf(stderr, "moni_hip: %s: not a liftidx file\n", ldx_path); return rc; }
        std::vector<uint8_t> text;
        if (text_path) {
            if (!refio::read_file(text_path, text)) return MONI_EIO;
            if (text.size() != M.n - 1) { fprintf(stderr, "moni_hip: %s holds %zu bytes, the BWT %llu\n", text_path, text.size(), (unsigned long long)M.n); return MONI_EINVAL; }
        }
        const uint64_t n_text = M.n - 1;
        refio::LdxFlat lf; lf.from(L);
        const uint64_t w = L.has_w ? L.w : 10;
        if (lf.seq_starts.empty() || (lf.seq_starts.back() != n_text && lf.seq_starts.back() + (w ? w - 1 : 0) != n_text)) {
            fprintf(stderr, "moni_hip: %s does not describe this text\n", ldx_path); return MONI_EINVAL;
        }
        moni_flat_index_t f; memset(&f, 0, sizeof f);
        f.n = M.n; f.r = M.r; f.F = M.F.data(); f.heads = M.heads.data(); f.starts = M.starts.data(); f.ssa = M.ssa.data(); f.esa = M.esa.data();
        f.thr = M.thr.data(); f.slcp = M.has_lcp ? M.slcp.data() : nullptr; f.text = text_path ? text.data() : nullptr;      // NULL: rebuilt from the BWT (moni_index_create)
        lf.fill(f, w);
        return moni_index_create(&f, device, out);
    } catch (const std::bad_alloc&) { return MONI_ENOMEM; }
}

int moni_ldx_rewrite(const char* in_path, const char* out_path, int with_w) {
    if (!in_path || !out_path) return MONI_EINVAL;
    try {
        refio::Ldx L;
        int rc = refio::load_ldx(in_path, L);
        if (rc) return rc;
        if (with_w && !L.has_w) L.w = 10;          // the older layout has no field for it: the separator width of every build of the reference
        return refio::save_ldx(out_path, L, with_w != 0);
    } catch (const std::bad_alloc&) { return MONI_ENOMEM; }
}

int moni_ldx_write(const moni_flat_index_t* f, const char* path, int with_w) {
    if (!f || !path || !f->seq_starts || !f->seq_names || f->n_seq < 1) return MONI_EINVAL;
    if (f->lift_second && (!f->lift_len || !f->lift_ins_off || !f->lift_del_off)) return MONI_EINVAL;
    try {
        refio::Ldx L;
        L.u = f->seq_starts[f->n_seq] + 1; L.w = f->w; L.has_w = with_w != 0;
        L.starts.size = L.u; L.starts.ones.assign(f->seq_starts, f->seq_starts + f->n_seq + 1);
        const char* nm = f->seq_names;
        for (uint64_t i = 0; i < f->n_seq; ++i) { L.names.emplace_back(nm); nm += L.names.back().size() + 1; }
        L.lifts.resize(f->n_seq);
        for (uint64_t i = 0; i < f->n_seq; ++i) {
            refio::LiftSd& x = L.lifts[i];
            const uint64_t span = f->seq_starts[i + 1] - f->seq_starts[i];
            const uint64_t len = f->lift_second ? f->lift_len[i] : span - (span >= f->w ? f->w : 0);
            x.second = f->lift_second ? f->lift_second[i] : f->seq_starts[i];
            x.ins.size = x.del.size = x.snp.size = len;
            if (f->lift_second) {
                if (f->lift_ins) x.ins.ones.assign(f->lift_ins + f->lift_ins_off[i], f->lift_ins + f->lift_ins_off[i + 1]);
                if (f->lift_del) x.del.ones.assign(f->lift_del + f->lift_del_off[i], f->lift_del + f->lift_del_off[i + 1]);
            }
        }
        return refio::save_ldx(path, L, with_w != 0);
    } catch (const std::bad_alloc&) { return MONI_ENOMEM; }
}

int moni_ldx_info(const char* path, uint64_t* n_seq, uint64_t* u, uint64_t* w, int* has_w) {
    if (!path) return MONI_EINVAL;
    try {
        refio::Ldx L;
        int rc = refio::load_ldx(path, L);
        if (rc) return rc;
        if (n_seq) *n_seq = L.names.size();
        if (u) *u = L.u;
        if (w) *w = L.w;
        if (has_w) *has_w = L.has_w ? 1 : 0;
        return MONI_OK;
    } catch (const std::bad_alloc&) { return MONI_ENOMEM; }
}

int moni_ldx_lift_batch(const char* path, int device, const uint64_t* pos, uint64_t n, uint64_t* out) {
    if (!path || (n && (!pos || !out))) return MONI_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) return MONI_ENODEV;
    HIPCHK(hipSetDevice(device));
    try {
    refio::Ldx L;
    int rc = refio::load_ldx(path, L);
    if (rc) return rc;
    refio::LdxFlat F; F.from(L);
    moni_flat_index_t f; memset(&f, 0, sizeof f);
    F.fill(f, L.has_w ? L.w : 10);
    f.n = F.seq_starts.back() + 1;
    LiftTables lt; std::string err;
    if ((rc = lt.build(f, err))) { fprintf(stderr, "moni_hip: %s\n", err.c_str()); return rc; }
    uint64_t bytes = 0;
    moni_lift_seq_t* d_seqs = nullptr; moni_lift_run_t* d_runs = nullptr; uint64_t *d_pdir = nullptr, *d_starts = nullptr, *d_io = nullptr;
    auto done = [&](int code) { void* ps[] = {d_seqs, d_runs, d_pdir, d_starts, d_io}; for (void* p : ps) if (p) (void)hipFree(p); return code; };
    if ((rc = upload(&d_seqs, lt.seqs, bytes)) || (rc = upload(&d_runs, lt.runs, bytes)) || (rc = upload(&d_pdir, lt.pdir, bytes)) || (rc = upload(&d_starts, F.seq_starts, bytes))) return done(rc);
    if (!n) return done(MONI_OK);
    if (hipMalloc((void**)&d_io, 2 * n * 8) != hipSuccess) return done(MONI_ENOMEM);
    if (hipMemcpy(d_io, pos, n * 8, hipMemcpyHostToDevice) != hipSuccess) return done(MONI_ENODEV);
    ac_params_t P; memset(&P, 0, sizeof P);
    P.n_text = f.n - 1; P.n_seq = (uint32_t)f.n_seq; P.seq_starts = d_starts; P.lift_seqs = d_seqs; P.lift_runs = d_runs; P.pdir = d_pdir;
    hipLaunchKernelGGL(lift_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, P, (const uint64_t*)d_io, n, d_io + n);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, d_io + n, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return done(MONI_ENODEV);
    return done(MONI_OK);
    } catch (const std::bad_alloc&) { return MONI_ENOMEM; }
}

int moni_sam_header(const moni_index_t* I, char** sam, uint64_t* sam_len) {
    if (!I || !sam || !sam_len) return MONI_EINVAL;
    const std::string h = I->hix.sam_header();
    *sam = (char*)malloc(h.size() + 1);
    if (!*sam) return MONI_ENOMEM;
    memcpy(*sam, h.data(), h.size() + 1);
    *sam_len = h.size();
    return MONI_OK;
}

#include "pe_api.inc"

}  // extern "C"
