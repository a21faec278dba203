// HIP kernels (gfx950) for the seeding half of the hot path; the per-lane logic is in seed_core.h.
//
// Mapping: one LANE per (read, strand) for the LF / MEM loops and one lane per MEM for the phi walks.
// Every step is a dependent random access into a multi-GB table, so the only parallelism is across
// reads; 64 independent reads per wavefront keep 64 row fetches in flight per wave instruction
// (a wave-per-read mapping keeps one), and the per-step state is a handful of registers so every
// SIMD runs 8 waves.  Tables are laid out so that a step reads one or two adjacent 16-byte rows
// (layout.h).  Byte tables live in LDS.  Integer/indexing work only: no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "seed_core.h"

#define MS_BLOCK 256
#ifndef MS_CHAINS
#define MS_CHAINS 2          // independent tasks interleaved per lane in ms_lf_kernel (both strands of one read)
#endif

__device__ __forceinline__ void load_tables(lds_tables_t& L, const moni_tables_t* __restrict__ T, const moni_consts_t& K) {
    if (threadIdx.x < MONI_MAX_SIGMA) {
        L.rec_base[threadIdx.x] = K.rec_base[threadIdx.x];
        L.rec_cnt[threadIdx.x] = K.rec_cnt[threadIdx.x];
        L.hot_slot[threadIdx.x] = K.hot_slot[threadIdx.x];
    }
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        L.code[i] = T->code[i];
        L.compl_tab[i] = T->compl_tab[i];
        L.c2[i] = base_acgt((uint32_t)i) ? (uint8_t)base2((uint32_t)i) : (uint8_t)4;
        L.abs_run[i] = T->abs_run[i];
        L.abs_pos[i] = T->abs_pos[i];
    }
    __syncthreads();
}

__device__ __forceinline__ void wave_add(unsigned long long v, unsigned long long* dst) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst, v);
}

extern "C" __global__ void __launch_bounds__(MS_BLOCK)
pack_kernel(const moni_consts_t K, const moni_tables_t* __restrict__ T, const uint8_t* __restrict__ seq, const uint64_t* __restrict__ offs, const moni_u64x2* __restrict__ blk,
            uint64_t n_tasks, uint64_t* __restrict__ pat, uint8_t* __restrict__ pflag) {
    __shared__ lds_tables_t L;
    load_tables(L, T, K);
    const uint64_t task = (uint64_t)blockIdx.x * MS_BLOCK + threadIdx.x;
    if (task < n_tasks) pack_task(L, seq, offs, blk, task, pat, pflag);
}

// NCH independent tasks per lane (see ms_task); MINW = minimum waves per SIMD the register allocator must leave room for.
template <int NCH, int MINW>
__global__ void __launch_bounds__(MS_BLOCK, MINW)
ms_lf_kernel(const moni_consts_t K, const moni_tables_t* __restrict__ T, const moni_row_t* __restrict__ rows,
             const moni_frow_t* __restrict__ frows, const uint32_t* __restrict__ cr, const moni_rec_t* __restrict__ recs,
             const uint64_t* __restrict__ pat, const uint64_t* __restrict__ offs, const moni_u64x2* __restrict__ blk, uint64_t n_tasks,
             uint64_t* __restrict__ ptr_out, unsigned long long* __restrict__ counters) {
    __shared__ lds_tables_t L;
    load_tables(L, T, K);
    const uint64_t task0 = ((uint64_t)blockIdx.x * MS_BLOCK + threadIdx.x) * NCH;
    unsigned long long n_steps = 0, n_jumps = 0;
    if (task0 < n_tasks) ms_task<NCH>(K, L, rows, frows, cr, recs, pat, offs, blk, n_tasks, task0, ptr_out, n_steps, n_jumps);
    wave_add(n_steps, &counters[0]);
    wave_add(n_jumps, &counters[1]);
}

// W: 64-bit pattern code words per lane in LDS (reads of up to 32 W bases take the 2-bit comparison; longer ones in the same launch compare bytes)
template <bool EMIT, int W>
__global__ void __launch_bounds__(MS_BLOCK)
mem_kernel(const moni_consts_t K, const uint8_t* __restrict__ text, const uint64_t* __restrict__ text2, const uint32_t* __restrict__ exc, uint32_t exc_sh, uint32_t exc_words,
           const uint64_t* __restrict__ pat, const uint64_t* __restrict__ offs, const moni_u64x2* __restrict__ blk, uint64_t n_tasks,
           const uint64_t* __restrict__ ptr, uint32_t min_len, uint32_t split_on,
           uint32_t* __restrict__ cnt_m, uint32_t* __restrict__ cnt_s,
           const uint64_t* __restrict__ read_mem_off, moni_mem_t* __restrict__ mems, uint32_t* __restrict__ aux,
           moni_u64x2* __restrict__ slots, unsigned long long* __restrict__ counters) {
    __shared__ uint64_t PW[W][MS_BLOCK];                 // [word][lane]: a lane's words lie in its own banks whatever word it reads
    __shared__ uint32_t EX[MONI_EXC_BITS / 32];
    for (uint32_t i = threadIdx.x; i < exc_words && i < MONI_EXC_BITS / 32; i += MS_BLOCK) EX[i] = exc[i];
    __syncthreads();
    mem_fast_t F;
    F.text2 = text2; F.exc = EX; F.exc_sh = exc_sh; F.pw = &PW[0][threadIdx.x]; F.pw_stride = MS_BLOCK; F.pw_words = W;
    const uint64_t task = (uint64_t)blockIdx.x * MS_BLOCK + threadIdx.x;
    unsigned long long n_cmp = 0;
    if (task < n_tasks)
        mem_task<EMIT>(K, F, text, pat, offs, blk, task, ptr, min_len, split_on, cnt_m, cnt_s, read_mem_off, mems, aux, slots, n_cmp);
    if (!EMIT) wave_add(n_cmp, &counters[3]);
}

// the 2-bit text and the bitmap of its blocks that hold a byte outside A / C / G / T (seed_core.h: text2_word), one thread per word; once per index
__global__ void __launch_bounds__(256) text2_build_kernel(const uint8_t* __restrict__ text, uint64_t n_text, uint64_t n_words, uint64_t* __restrict__ text2,
                                                          uint32_t* __restrict__ exc, uint32_t exc_sh) {
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= n_words) return;
    bool bad;
    text2[w] = text2_word(text, n_text, w, bad);
    if (bad) { const uint64_t b = (32 * w) >> exc_sh; atomicOr(&exc[b >> 5], 1u << (b & 31u)); }
}

// per-read final MEM count (orig + 2 * split)
extern "C" __global__ void read_totals_kernel(const uint32_t* __restrict__ cnt_m, const uint32_t* __restrict__ cnt_s,
                                              uint64_t n_reads, uint64_t* __restrict__ tot) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads) tot[r] = (uint64_t)cnt_m[2 * r] + cnt_m[2 * r + 1] + 2ull * ((uint64_t)cnt_s[2 * r] + cnt_s[2 * r + 1]);
    if (r == n_reads) tot[r] = 0;
}

extern "C" __global__ void phi_batch_kernel(const moni_consts_t K, phi_tab_t P, const uint64_t* __restrict__ pos, uint64_t n,
                                            uint64_t* __restrict__ out_pos, uint64_t* __restrict__ out_lcp) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { uint64_t a, b; phi_step(P, K, pos[t], a, b); out_pos[t] = a; out_lcp[t] = b; }
}

template <bool FILL>
__global__ void __launch_bounds__(MS_BLOCK)
occ_kernel(const moni_consts_t K, occ_args_t A) {
    const uint64_t g = (uint64_t)blockIdx.x * MS_BLOCK + threadIdx.x;
    unsigned long long phi_steps = 0;
    if (g < A.n_mems) occ_task<FILL>(K, A, g, phi_steps);
    if (!FILL) wave_add(phi_steps, &A.counters[2]);
}

// gather occ_cnt into a u64 array for the scan, and scatter the scanned offsets back
extern "C" __global__ void occ_cnt_gather_kernel(const moni_mem_t* __restrict__ mems, uint64_t n, uint64_t* __restrict__ cnt) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) cnt[g] = mems[g].occ_cnt;
    if (g == n) cnt[g] = 0;
}
extern "C" __global__ void occ_off_scatter_kernel(moni_mem_t* __restrict__ mems, uint64_t n, const uint64_t* __restrict__ off) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) mems[g].occ_off = off[g];
}


// Matching-statistics lengths of the forward strand (src/matching_statistics.cpp:242-256, src/mems.cpp:241-258: the loop both legacy
// front ends run over ms.query's pointers): lane = read, lens[offs[read] - offs[0] + i] = l at read offset i.
__global__ void __launch_bounds__(MS_BLOCK)
ms_len_kernel(const moni_consts_t K, const moni_tables_t* __restrict__ T, const uint8_t* __restrict__ text, const uint64_t* __restrict__ pat,
              const uint64_t* __restrict__ offs, const moni_u64x2* __restrict__ blk, uint64_t n_reads, const uint64_t* __restrict__ ptr, uint32_t* __restrict__ lens) {
    __shared__ lds_tables_t L;
    load_tables(L, T, K);
    const uint64_t read = (uint64_t)blockIdx.x * MS_BLOCK + threadIdx.x;
    if (read >= n_reads) return;
    const uint64_t task = 2 * read;
    const uint64_t pb = ws_pat_base(blk, task), qb = ws_ptr_base(blk, task);
    const uint64_t off = offs[read];
    const uint32_t m = (uint32_t)(offs[read + 1] - off);
    const uint64_t n = K.n_text;
    uint64_t l = 0, prev_pos_plus_one = n + 1;
    pat_cache_t pc; pc.w = 0xFFFFFFFFu; pc.word = 0;
    text_cache_t tc; tc.w = ~0ull; tc.word = 0;
    for (uint32_t i = 0; i < m; ++i) {
        const uint64_t pos = ptr[qb + (uint64_t)(m - 1 - i) * 64u];
        while (pos != prev_pos_plus_one && (i + l) < m && (pos + l) < n) {
            if (pat_byte(pat, pb, m, (uint32_t)(i + l), pc) != text_byte(text, pos + l, tc)) break;
            ++l;
        }
        lens[off - offs[0] + i] = (uint32_t)l;
        l = (l == 0 ? 0 : (l - 1));
        prev_pos_plus_one = pos + 1;
    }
}

// per-seed largest / smallest per-genome occurrence count of slots [g0, g1) (`-c`, genome_task)
__global__ void __launch_bounds__(MS_BLOCK) genome_kernel(const moni_consts_t K, occ_args_t A, uint64_t g0, uint64_t g1, uint32_t* __restrict__ rows, uint64_t* __restrict__ hi_lo) {
    const uint64_t g = g0 + (uint64_t)blockIdx.x * MS_BLOCK + threadIdx.x;
    if (g < g1) genome_task(K, A, g, g0, rows, hi_lo);
}
