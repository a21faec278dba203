// Per-lane logic of the seeding kernels, written once and compiled twice: as __device__ code inside
// seed_kernels.hip (the product), and as plain host C++ by tests/host_sim.cpp, which replays the
// same functions lane by lane over the host copy of the index image so that the layout and the step
// logic can be checked against the oracle without a GPU.  The host build exists only under tests/.
//
//   ms_task    ms_pointers::_query            include/ms/moni.hpp:568-624
//   mem_task   seed_finder::find_mems         include/aligner/seed_finder.hpp:126-166
//   occ_task   seed_finder::populate_seed(s)  include/aligner/seed_finder.hpp:169-343, 377-393
//              + moni_lcp::Phi_lcp/Phi_inv_lcp include/aligner/moni_lcp.hpp:230-272
#pragma once
#include <stdint.h>

#include "../../include/moni_hip.h"
#include "layout.h"

#if defined(__HIPCC__)
#define MONI_HD __host__ __device__ __forceinline__
#else
#define MONI_HD inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define MONI_ATOMIC_INC_U32(p) atomicAdd((p), 1u)
#define MONI_FLAG_SET(p) atomicExch((p), 1u)
#else
#define MONI_ATOMIC_INC_U32(p) ((*(p))++)
#define MONI_FLAG_SET(p) (*(p) = 1u)
#endif

#define MONI_MEM_SLOTS 4u
struct alignas(16) moni_u64x2 { uint64_t x, y; };
struct alignas(32) moni_u64x4 { uint64_t x, y, z, w; };

struct lds_tables_t {
    uint8_t code[256];
    uint8_t compl_tab[256];
    uint8_t c2[256];                       // byte -> 2-bit code of mem_task's comparison (base2), 4 for a byte that is not A / C / G / T
    uint32_t abs_run[256];
    uint64_t abs_pos[256];
    uint32_t rec_base[MONI_MAX_SIGMA];     // copies of the per-code constants, so that a divergent code index is an LDS read
    uint32_t rec_cnt[MONI_MAX_SIGMA];
    uint32_t hot_slot[MONI_MAX_SIGMA];
};

MONI_HD uint64_t row_start(const moni_row_t& x) { return x.w0 & MONI_POS_MASK; }
MONI_HD uint32_t row_head(const moni_row_t& x) { return (uint32_t)(x.w0 >> 40) & 15u; }
MONI_HD uint32_t row_len(const moni_row_t& x) { return (uint32_t)(x.w0 >> 52); }           // saturates at 4095
MONI_HD uint64_t row_lfbase(const moni_row_t& x) { return x.w1 & MONI_POS_MASK; }
MONI_HD uint32_t row_dest(const moni_row_t& x) { return (uint32_t)(((x.w0 >> 44) & 0xFFu) << 24) | (uint32_t)(x.w1 >> 40); }

MONI_HD moni_row_t ld_row(const moni_row_t* __restrict__ rows, uint32_t k) {
    const moni_u64x4 v = *reinterpret_cast<const moni_u64x4*>(rows + k);           // one 32-byte aligned load
    moni_row_t x; x.w0 = v.x; x.w1 = v.y;
    x.hot_cr[0] = (uint32_t)v.z; x.hot_cr[1] = (uint32_t)(v.z >> 32); x.hot_cr[2] = (uint32_t)v.w; x.hot_cr[3] = (uint32_t)(v.w >> 32);
    return x;
}
// hot_cr[slot] without a runtime-indexed register array (which would live in scratch)
MONI_HD uint32_t row_hot(const moni_row_t& x, uint32_t slot) {
    return slot == 0 ? x.hot_cr[0] : slot == 1 ? x.hot_cr[1] : slot == 2 ? x.hot_cr[2] : x.hot_cr[3];
}
MONI_HD uint64_t ld_start(const moni_row_t* __restrict__ rows, uint32_t k) { return rows[k].w0 & MONI_POS_MASK; }

// Is pos inside run A (which starts at or before pos)?  The row carries min(len, 4095); only runs at
// least that long need the next row's start.
MONI_HD bool in_run(const moni_row_t* __restrict__ rows, uint32_t run, const moni_row_t& A, uint64_t pos) {
    const uint64_t off = pos - row_start(A);
    const uint32_t len = row_len(A);
    if (off < len) return true;
    if (len < MONI_ROW_LEN_SAT) return false;
    return pos < ld_start(rows, run + 1);
}

// Find the run with start[run] <= pos < start[run+1], starting from a guess.  The guess is the
// destination run of the LF mapping, so the answer is normally the guess or a close successor; a
// galloping search bounds the cost when a long run maps onto many short ones.
MONI_HD void settle_run(const moni_row_t* __restrict__ rows, uint64_t r, uint64_t pos, uint32_t& run, moni_row_t& A) {
    A = ld_row(rows, run);
    if (pos < row_start(A)) {          // a jump up lands on lfpos-1, possibly the run before the stored one
        uint32_t step = 1;
        uint32_t hi = run;             // start[hi] > pos
        uint32_t lo = run >= step ? run - step : 0;
        while (lo > 0 && ld_start(rows, lo) > pos) { hi = lo; step <<= 1; lo = lo >= step ? lo - step : 0; }
        while (hi - lo > 1) { uint32_t mid = lo + ((hi - lo) >> 1); if (ld_start(rows, mid) <= pos) lo = mid; else hi = mid; }
        run = lo; A = ld_row(rows, run);
        return;
    }
    int lin = 0;
    while (!in_run(rows, run, A, pos)) {
        if (++lin > 3) {
            uint32_t lo = run;         // start[lo] <= pos
            uint32_t step = 4;
            uint32_t hi = lo + step;
            const uint32_t top = (uint32_t)r + 1;   // start[r+1] = 2^40-1 > any pos
            while (true) { if (hi > top) hi = top; if (ld_start(rows, hi) > pos) break; lo = hi; step <<= 1; hi = lo + step; }
            while (hi - lo > 1) { uint32_t mid = lo + ((hi - lo) >> 1); if (ld_start(rows, mid) <= pos) lo = mid; else hi = mid; }
            run = lo; A = ld_row(rows, run);
            return;
        }
        ++run; A = ld_row(rows, run);
    }
}

// ------------------------------------------------------------------------------------------------
// Packed patterns: pat[ws_pat_base(task) + w * 64] holds the bytes the LF loop consumes in steps 8w .. 8w+7 of a task
// (step s reads pattern[m-1-s]; the reverse-complement strand is complemented here: aligner_ksw2.hpp:169-176,
// kpbseq.h:150-168).  One coalesced 8-byte load per lane every 8 steps replaces a byte load per step.
// ------------------------------------------------------------------------------------------------
// bytes a .. a+7 of a buffer that is 8-byte aligned and padded by 16 bytes: two aligned word loads instead of eight byte loads
// (64 lanes on 64 different reads make every byte load a separate request)
MONI_HD uint64_t load8_unaligned(const uint8_t* __restrict__ p, uint64_t a) {
    const uint64_t w = a & ~7ull;
    const uint32_t sh = (uint32_t)(a & 7u) * 8u;
    const uint64_t lo = *reinterpret_cast<const uint64_t*>(p + w);
    if (!sh) return lo;
    const uint64_t hi = *reinterpret_cast<const uint64_t*>(p + w + 8);
    return (lo >> sh) | (hi << (64u - sh));
}
MONI_HD uint64_t bswap64_(uint64_t v) {
    v = ((v & 0x00FF00FF00FF00FFull) << 8) | ((v >> 8) & 0x00FF00FF00FF00FFull);
    v = ((v & 0x0000FFFF0000FFFFull) << 16) | ((v >> 16) & 0x0000FFFF0000FFFFull);
    return (v << 32) | (v >> 32);
}

// Workspace layout of the per-step arrays (packed patterns, MS pointers): tasks are taken 64 at a time (a wavefront's worth: 32 reads x 2
// strands); block b holds its tasks' values step-major, `base + step * 64 + (task & 63)`, over as many steps as its LONGEST read has.
// Lanes of a wave touch consecutive words at every step, and one long read costs 64 x its length in its own block only - not
// n_tasks x its length as a layout strided by the batch's longest read does.  blk[b] = {first pointer word, first pattern word} of block b
// (exclusive prefix sums, built by the host when the batch is uploaded); blk[n_blocks] = the totals.
MONI_HD uint64_t ws_ptr_base(const moni_u64x2* __restrict__ blk, uint64_t task) { return blk[task >> 6].x + (task & 63u); }
MONI_HD uint64_t ws_pat_base(const moni_u64x2* __restrict__ blk, uint64_t task) { return blk[task >> 6].y + (task & 63u); }

// ------------------------------------------------------------------------------------------------
// 2-bit forms for mem_task's comparison loop (seed_finder.hpp:134-150 compares one byte at a time through the SLP): the text as 32 bases per
// 64-bit word (text2: 200 MB for 800 M bases - the random accesses of the loop then hit a table that the 256 MB Infinity Cache can hold), and
// every task's strand-resolved pattern likewise, in READ order (base qi of the pattern in word qi >> 5, bits 2 (qi & 31) ..).  Only A / C / G / T
// have a code; a byte outside them reads as code 0 and is marked: text bytes in a bitmap with one bit per block of 2^exc_sh positions
// (separators, N runs: a comparison that touches a marked block is redone byte by byte), pattern bytes in a mask word beside the code word
// (the 2-bit comparison stops in front of them).  Results are those of the byte comparison in every case.
// ------------------------------------------------------------------------------------------------
MONI_HD uint32_t base2(uint32_t b) { return (b >> 1) & 3u; }              // 'A' 0, 'C' 1, 'T' 2, 'G' 3
MONI_HD bool base_acgt(uint32_t b) { return b == 'A' || b == 'C' || b == 'G' || b == 'T'; }
MONI_HD uint32_t ctz64_(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }
// word w of the 2-bit text: bases 32 w .. 32 w + 31 (beyond n_text: 0); any_bad: one of them is not A / C / G / T
MONI_HD uint64_t text2_word(const uint8_t* __restrict__ text, uint64_t n_text, uint64_t w, bool& any_bad) {
    uint64_t word = 0;
    any_bad = false;
    for (uint32_t k = 0; k < 32; ++k) {
        const uint64_t p = 32 * w + k;
        if (p >= n_text) break;
        const uint32_t b = text[p];
        if (base_acgt(b)) word |= (uint64_t)base2(b) << (2 * k); else any_bad = true;
    }
    return word;
}
// the smallest block size 2^sh (sh >= 10) whose bitmap has at most `max_bits` bits for a text of n_text bases (+ 1: position n_text itself is asked for)
MONI_HD uint32_t text2_exc_shift(uint64_t n_text, uint64_t max_bits) { uint32_t sh = 10; while (((n_text >> sh) + 1) > max_bits) ++sh; return sh; }
#define MONI_EXC_BITS 65536u              // 8 KB of LDS in mem_kernel

// the pattern workspace of a block of 64 tasks whose longest read has lb bases: [bytes, 8 per word][codes, 32 per word][masks, 32 per word], each [word][task & 63]
MONI_HD uint64_t ws_pat_words(uint64_t lb) { return (lb + 7) / 8 + 2 * ((lb + 31) / 32); }
MONI_HD uint64_t ws_block_len(const moni_u64x2* __restrict__ blk, uint64_t task) { return (blk[(task >> 6) + 1].x - blk[task >> 6].x) >> 6; }      // (a block's pointer words: 64 per step)

MONI_HD void pack_task(const lds_tables_t& L, const uint8_t* __restrict__ seq, const uint64_t* __restrict__ offs, const moni_u64x2* __restrict__ blk,
                       uint64_t task, uint64_t* __restrict__ pat, uint8_t* __restrict__ pflag = nullptr) {      // pflag[task]: the pattern holds a byte outside A / C / G / T
    const uint64_t read = task >> 1;
    const uint32_t strand = (uint32_t)task & 1u;
    const uint64_t off = offs[read];
    const uint32_t m = (uint32_t)(offs[read + 1] - off);
    const uint64_t pb = ws_pat_base(blk, task);
    const uint32_t n_words = (m + 7u) >> 3;               // the words this task's loops read
    for (uint32_t w = 0; w < n_words; ++w) {
        uint64_t word = 0;
        const uint32_t s0 = 8 * w;
        {
            const uint32_t nv = m - s0 < 8 ? m - s0 : 8;          // bytes of this word inside the pattern
            if (strand) {                                          // the read's bytes s0 .. s0 + 7, complemented
                const uint64_t v = load8_unaligned(seq, off + s0);
                for (uint32_t j = 0; j < nv; ++j) word |= (uint64_t)L.compl_tab[(uint32_t)(v >> (8 * j)) & 0xFFu] << (8 * j);
            } else if (nv == 8) {                                  // the read's bytes m-1-s0 down to m-8-s0
                word = bswap64_(load8_unaligned(seq, off + m - 8 - s0));
            } else {
                for (uint32_t j = 0; j < nv; ++j) word |= (uint64_t)seq[off + (m - 1 - s0 - j)] << (8 * j);
            }
        }
        pat[pb + (uint64_t)w * 64u] = word;
    }
    // the same pattern in read order, 2 bits per base: pattern[qi] = seq[off + qi] (forward strand) or the complement of seq[off + m - 1 - qi]
    const uint64_t lb = ws_block_len(blk, task);
    const uint64_t cb = pb + 64u * ((lb + 7) / 8), mb = cb + 64u * ((lb + 31) / 32);
    const uint32_t n2 = (m + 31u) >> 5;
    uint64_t any_mask = 0;
    for (uint32_t w = 0; w < n2; ++w) {
        uint64_t codes = 0, mask = 0;
        for (uint32_t g = 0; g < 4; ++g) {
            const uint32_t q0 = 32 * w + 8 * g;
            if (q0 >= m) break;
            const uint32_t nv = m - q0 < 8 ? m - q0 : 8;
            uint64_t v = 0;                                        // pattern[q0 + j] in byte j
            if (!strand) {
                if (nv == 8) v = load8_unaligned(seq, off + q0);
                else for (uint32_t j = 0; j < nv; ++j) v |= (uint64_t)seq[off + q0 + j] << (8 * j);
            } else {
                uint64_t u = 0;                                    // seq[off + m - 1 - q0 - j] in byte j
                if (nv == 8) u = bswap64_(load8_unaligned(seq, off + m - 8 - q0));
                else for (uint32_t j = 0; j < nv; ++j) u |= (uint64_t)seq[off + (m - 1 - q0 - j)] << (8 * j);
                for (uint32_t j = 0; j < nv; ++j) v |= (uint64_t)L.compl_tab[(uint32_t)(u >> (8 * j)) & 0xFFu] << (8 * j);
            }
            uint32_t c16 = 0, b16 = 0;
            for (uint32_t j = 0; j < nv; ++j) { const uint32_t t = L.c2[(uint32_t)(v >> (8 * j)) & 0xFFu]; c16 |= (t & 3u) << (2 * j); b16 |= (t >> 2) << (2 * j); }
            codes |= (uint64_t)c16 << (16 * g); mask |= (uint64_t)b16 << (16 * g);
        }
        pat[cb + (uint64_t)w * 64u] = codes;
        pat[mb + (uint64_t)w * 64u] = mask;
        any_mask |= mask;
    }
    if (pflag) pflag[task] = any_mask ? 1 : 0;
}

// byte qi of the strand-oriented pattern (= the byte consumed at step m-1-qi), through a one-word register cache
struct pat_cache_t { uint64_t word; uint32_t w; };
MONI_HD uint8_t pat_byte(const uint64_t* __restrict__ pat, uint64_t pb, uint32_t m, uint32_t qi, pat_cache_t& c) {      // pb = ws_pat_base(blk, task)
    const uint32_t s = m - 1 - qi;
    const uint32_t w = s >> 3;
    if (w != c.w) { c.word = pat[pb + (uint64_t)w * 64u]; c.w = w; }
    return (uint8_t)(c.word >> (8 * (s & 7)));
}
// byte a of the text through a one-word register cache (text is allocated 8-byte aligned and padded)
struct text_cache_t { uint64_t word; uint64_t w; };
MONI_HD uint8_t text_byte(const uint8_t* __restrict__ text, uint64_t a, text_cache_t& c) {
    const uint64_t w = a >> 3;
    if (w != c.w) { c.word = *reinterpret_cast<const uint64_t*>(text + (w << 3)); c.w = w; }
    return (uint8_t)(c.word >> (8 * (a & 7)));
}

// ------------------------------------------------------------------------------------------------
// ms_task: pointers[ws_ptr_base(task) + s * 64] = sample after step s, i.e. ms_pointers[m-1-s];  task = 2*read + strand
// ------------------------------------------------------------------------------------------------
// State of one task's LF loop.  In the common case it is (run, off): the position is offset `off` inside run `run`
// (MONI_OFF_END = last position of the run), and a step reads exactly one 64-byte fast row.  After a step that went
// through the general path the state is an absolute BWT position (abs = true) and is re-anchored at the next step.
struct ms_state_t {
    uint64_t pos, sample, word;
    uint32_t run, off, m;
    bool abs;
};

// The general path of one step: absolute position, rows / cr / recs (moni.hpp:589-618).  c is a symbol of the BWT.
MONI_HD void ms_step_general(const moni_consts_t& K, const lds_tables_t& L, const moni_row_t* __restrict__ rows,
                             const uint32_t* __restrict__ cr, const moni_rec_t* __restrict__ recs, uint32_t c, ms_state_t& S,
                             unsigned long long& n_jumps) {
    moni_row_t A;
    settle_run(rows, K.r, S.pos, S.run, A);
    if (row_head(A) == c) {                                  // bwt[pos] == c  (moni.hpp:589-594); the sentinel head never matches
        S.sample--;
        S.pos = row_lfbase(A) + (S.pos - row_start(A));
        S.run = row_dest(A);
    } else {                                                 // threshold jump (moni.hpp:595-618)
        ++n_jumps;
        const uint32_t hs = L.hot_slot[c];
        const uint32_t j = hs < 4 ? row_hot(A, hs) : cr[(uint64_t)S.run * K.sigma + c];
        const moni_u64x4 rv = *reinterpret_cast<const moni_u64x4*>(recs + L.rec_base[c] + j);
        const uint64_t thr = rv.x & MONI_POS_MASK;
        const uint32_t d = (uint32_t)((rv.x >> 40) << 24) | (uint32_t)(rv.y >> 40);
        // rnk_c.first > thresholds.rank(pos+1, c)  <=>  j >= 1 and (no c-run below, or pos < thr_j)
        const bool up = j > 0 && (j == L.rec_cnt[c] || S.pos < thr);
        if (up) { S.sample = rv.z; S.pos = rv.w - 1; }
        else { S.sample = rv.y & MONI_POS_MASK; S.pos = rv.w; }
        S.run = d;
    }
    S.abs = true;
}

// -DMONI_MS_ATTR=1 / 2 (profiles/ms_attr.sh; never in the product build): the jump counter's bits 28.. count what a step reads beyond its own fast row -
// 1: the next-run walks (the LF image lies past the destination run: one more fast row each), 2: the entries into the general path (32-byte row, c-run
// record, ...).  J itself stays in bits 0..27.
#if defined(MONI_MS_ATTR) && MONI_MS_ATTR == 1
#define MS_ATTR_WALK(nj) ((nj) += (1ull << 28))
#define MS_ATTR_GEN(nj)
#elif defined(MONI_MS_ATTR) && MONI_MS_ATTR == 2
#define MS_ATTR_WALK(nj)
#define MS_ATTR_GEN(nj) ((nj) += (1ull << 28))
#else
#define MS_ATTR_WALK(nj)
#define MS_ATTR_GEN(nj)
#endif
// One LF step of one task for symbol code c (moni.hpp:579-621).
MONI_HD void ms_step(const moni_consts_t& K, const lds_tables_t& L, const moni_row_t* __restrict__ rows,
                     const moni_frow_t* __restrict__ frows, const uint32_t* __restrict__ cr, const moni_rec_t* __restrict__ recs,
                     uint32_t c, ms_state_t& S, unsigned long long& n_jumps) {
    if (S.abs) {                                             // re-anchor an absolute position: (run, off)
        moni_row_t A;
        settle_run(rows, K.r, S.pos, S.run, A);
        S.off = (uint32_t)(S.pos - row_start(A));            // < 2^32 unless the run is longer, in which case the row is not "ok" anyway
        if (S.pos - row_start(A) >= MONI_ROW_LEN_SAT) { MS_ATTR_GEN(n_jumps); ms_step_general(K, L, rows, cr, recs, c, S, n_jumps); return; }
        S.abs = false;
    }
    const uint32_t hc = L.hot_slot[c];
    while (true) {
        const moni_frow_t* __restrict__ fr = frows + S.run;
        const moni_u64x2 q0 = *reinterpret_cast<const moni_u64x2*>(&fr->w[0]);   // w0, w1
        const uint64_t w0 = q0.x;
        const uint32_t len = (uint32_t)w0 & 0xFFFu;
        if (!((w0 >> 58) & 1u) || hc >= 4) {                 // general path: absolute position from the 32-byte row
            const moni_row_t A = ld_row(rows, S.run);
            if (S.off == MONI_OFF_END) S.pos = ld_start(rows, S.run + 1) - 1;
            else S.pos = row_start(A) + S.off;
            MS_ATTR_GEN(n_jumps);
            ms_step_general(K, L, rows, cr, recs, c, S, n_jumps);
            return;
        }
        if (S.off == MONI_OFF_END) S.off = len - 1;
        if (S.off >= len) { S.off -= len; ++S.run; MS_ATTR_WALK(n_jumps); continue; }          // the LF image ran past the destination run: next run
        const uint32_t hh = (uint32_t)(w0 >> 56) & 3u;
        if (hc == hh) {                                      // bwt[pos] == c
            S.sample--;
            S.off += (uint32_t)(w0 >> 12) & 0xFFFu;
            S.run = (uint32_t)(w0 >> 24);
            return;
        }
        ++n_jumps;
        const uint32_t sl = (hc - hh - 1u) & 3u;             // 0..2
        const moni_u64x2 q1 = *reinterpret_cast<const moni_u64x2*>(&fr->w[2]);   // w2, w3
        const moni_u64x2 q2 = *reinterpret_cast<const moni_u64x2*>(&fr->w[4]);   // w4, w5
        const moni_u64x2 q3 = *reinterpret_cast<const moni_u64x2*>(&fr->w[6]);   // w6, w7
        const uint64_t ws = sl == 0 ? q0.y : sl == 1 ? q1.x : q1.y;
        const uint32_t thr_off = (uint32_t)ws & 0xFFFu;
        const uint32_t sdoff = (uint32_t)(ws >> 12) & 0xFFFu;
        const uint32_t sdest = (uint32_t)(ws >> 24);
        if (S.off < thr_off) {                               // jump up: last position of the previous c-run
            const uint64_t lo = sl == 0 ? (q2.y >> 32) : sl == 1 ? (q3.x & 0xFFFFFFFFull) : (q3.x >> 32);
            const uint64_t hi = (q3.y >> (8 * sl)) & 0xFFull;
            S.sample = lo | (hi << 32);
            if (sdoff == 0) { S.run = sdest - 1; S.off = MONI_OFF_END; }
            else { S.run = sdest; S.off = sdoff - 1; }
        } else {                                             // jump down: first position of the next c-run
            const uint64_t lo = sl == 0 ? (q2.x & 0xFFFFFFFFull) : sl == 1 ? (q2.x >> 32) : (q2.y & 0xFFFFFFFFull);
            S.sample = lo | ((ws >> 56) << 32);
            S.run = sdest; S.off = sdoff;
        }
        return;
    }
}

// One lane runs NCH tasks (task0 .. task0+NCH-1; with NCH = 2 the two strands of one read).
template <int NCH>
MONI_HD void ms_task(const moni_consts_t& K, const lds_tables_t& L, const moni_row_t* __restrict__ rows,
                     const moni_frow_t* __restrict__ frows, const uint32_t* __restrict__ cr, const moni_rec_t* __restrict__ recs,
                     const uint64_t* __restrict__ pat, const uint64_t* __restrict__ offs, const moni_u64x2* __restrict__ blk, uint64_t n_tasks, uint64_t task0,
                     uint64_t* __restrict__ ptr_out, unsigned long long& n_steps, unsigned long long& n_jumps) {
    ms_state_t S[NCH];
    uint64_t pb[NCH], qb[NCH];
    uint32_t m_max = 0;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const uint64_t task = task0 + k;
        S[k].m = 0;
        pb[k] = qb[k] = 0;
        if (task < n_tasks) { const uint64_t read = task >> 1; S[k].m = (uint32_t)(offs[read + 1] - offs[read]); pb[k] = ws_pat_base(blk, task); qb[k] = ws_ptr_base(blk, task); }
        m_max = S[k].m > m_max ? S[k].m : m_max;
        S[k].run = (uint32_t)K.r - 1; S[k].pos = K.n - 1; S[k].abs = true; S[k].off = 0;     // start with the empty string
        S[k].sample = K.last_run_sample; S[k].word = 0;
        n_steps += S[k].m;
    }
    for (uint32_t s = 0; s < m_max; ++s) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (s >= S[k].m) continue;
            // pattern[m-1-s], already strand-resolved by pack_task
            if ((s & 7u) == 0) S[k].word = pat[pb[k] + (uint64_t)(s >> 3) * 64u];
            const uint32_t raw = (uint32_t)S[k].word & 0xFFu;
            S[k].word >>= 8;
            const uint32_t c = L.code[raw];
            if (c == MONI_CODE_ABSENT) {                      // n_c == 0   (moni.hpp:583-588)
                S[k].sample = 0;
                S[k].pos = L.abs_pos[raw];
                S[k].run = L.abs_run[raw];
                S[k].abs = true;
            } else {
                ms_step(K, L, rows, frows, cr, recs, c, S[k], n_jumps);
            }
            ptr_out[qb[k] + (uint64_t)s * 64u] = S[k].sample;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// mem_task<EMIT>: the forward text-comparison loop of find_mems.
//   EMIT = false: cnt_m[task] = number of MEMs, cnt_s[task] = number of MEMs that will be split in halves
//   EMIT = true : writes the MEM (and the fixed fields of its two halves) to its final slot.
// Final order per read (seed_finder.hpp:311-318 with aligner_ksw2.hpp:333-337): forward MEMs, reverse-
// complement MEMs, then for every MEM in that order [left half, right half] if len >= 2*min_len.
// aux[g]: 0xFFFFFFFF plain MEM, 0xFFFFFFFE left half, 0xFFFFFFFD right half, else offset of the MEM's halves
// from the read's first slot.
// ------------------------------------------------------------------------------------------------
// what the 2-bit comparison reads (text2 == nullptr: every comparison byte by byte)
struct mem_fast_t {
    const uint64_t* text2;            // 32 bases per word (text2_word)
    const uint32_t* exc;              // bit b: text block [b << exc_sh, (b + 1) << exc_sh) holds a byte that is not A / C / G / T (the kernel's copy is in LDS)
    uint32_t exc_sh;
    uint64_t* pw;                     // this lane's pattern code words, word w at pw[w * pw_stride] (LDS); filled here from the pattern workspace
    uint32_t pw_stride, pw_words;     // tasks of up to 32 * pw_words bases take the 2-bit comparison
};

template <bool EMIT>
MONI_HD void mem_task(const moni_consts_t& K, const mem_fast_t& F, const uint8_t* __restrict__ text,
                      const uint64_t* __restrict__ pat, const uint64_t* __restrict__ offs, const moni_u64x2* __restrict__ blk, uint64_t task,
                      const uint64_t* __restrict__ ptr, uint32_t min_len, uint32_t split_on, uint32_t* __restrict__ cnt_m,
                      uint32_t* __restrict__ cnt_s, const uint64_t* __restrict__ read_mem_off, moni_mem_t* __restrict__ mems,
                      uint32_t* __restrict__ aux, moni_u64x2* __restrict__ slots, unsigned long long& n_cmp) {
    const uint64_t read = task >> 1;
    const uint32_t strand = (uint32_t)task & 1u;
    const uint64_t off = offs[read];
    const uint32_t m = (uint32_t)(offs[read + 1] - off);
    const uint64_t n = K.n_text;                       // seed_finder::n = ra.getLen()
    uint64_t l = 0, pl = 0, n_Ns = 0;
    uint64_t prev_pos_plus_one = n + 1;
    uint32_t km = 0, ks = 0;
    uint64_t base = 0, k_read = 0, j0 = 0, s0 = 0;
    if (EMIT) {
        base = read_mem_off[read];
        k_read = (uint64_t)cnt_m[2 * read] + cnt_m[2 * read + 1];
        if (strand) { j0 = cnt_m[2 * read]; s0 = cnt_s[2 * read]; }
    }
    // one MEM found at read offset i with pointer pos and length l
    auto found = [&](uint64_t pos, uint64_t l, uint32_t i) {
        const bool split = split_on && l >= ((uint64_t)min_len << 1);
        if (EMIT) {
            const uint64_t g = base + j0 + km;
            moni_mem_t M;
            M.pos = pos; M.len = (uint32_t)l; M.idx = i; M.rpos = (uint32_t)(i + l - 1); M.mate = strand ? 2u : 0u;
            M.total_occ = 0; M.num_filtered = 0; M.occ_off = 0; M.occ_cnt = 0; M.read = (uint32_t)read;
            mems[g] = M;
            if (split) {
                const uint64_t hb = base + k_read + 2 * (s0 + ks);
                const uint32_t ll = (uint32_t)(l >> 1);
                moni_mem_t B = M;                    // left half: pos is the parent's upper suffix (occ_task)
                B.pos = 0; B.len = ll; B.rpos = (uint32_t)((i + l - 1) - l + ll);
                mems[hb] = B;
                moni_mem_t C = M;                    // right half (seed_finder.hpp:296-298)
                C.pos = pos + ll; C.len = (uint32_t)(l - ll); C.idx = i + ll;
                mems[hb + 1] = C;
                aux[g] = (uint32_t)(hb - base);
                aux[hb] = 0xFFFFFFFEu;
                aux[hb + 1] = 0xFFFFFFFDu;
            } else {
                aux[g] = 0xFFFFFFFFu;
            }
        } else if (km < MONI_MEM_SLOTS) {            // remember the first few MEMs so the emit pass need not redo the text walk
            moni_u64x2 sl; sl.x = pos; sl.y = (l << 32) | i;
            slots[task * MONI_MEM_SLOTS + km] = sl;
        }
        ++km;
        if (split) ++ks;
    };
    if (EMIT && cnt_m[task] <= MONI_MEM_SLOTS) {
        const uint32_t cnt = cnt_m[task];
        for (uint32_t k = 0; k < cnt; ++k) {
            const moni_u64x2 sl = slots[task * MONI_MEM_SLOTS + k];
            found(sl.x, sl.y >> 32, (uint32_t)sl.y);
        }
        return;
    }
    pat_cache_t pc; pc.w = 0xFFFFFFFFu; pc.word = 0;
    text_cache_t tc; tc.w = ~0ull; tc.word = 0;
    const uint64_t pb = ws_pat_base(blk, task), qb = ws_ptr_base(blk, task);
    // the pattern's code words into the lane's LDS column; has_inv: the pattern holds a byte that is not A / C / G / T (its mask words are then read
    // where they lie in the workspace)
    const bool fast = F.text2 != nullptr && m <= 32u * F.pw_words;
    uint64_t mb = 0;
    bool has_inv = false;
    if (fast) {
        const uint64_t lb = ws_block_len(blk, task);
        const uint64_t cb = pb + 64u * ((lb + 7) / 8);
        mb = cb + 64u * ((lb + 31) / 32);
        for (uint32_t w = 0; w < ((m + 31u) >> 5); ++w) { F.pw[w * F.pw_stride] = pat[cb + (uint64_t)w * 64u]; has_inv = has_inv || pat[mb + (uint64_t)w * 64u] != 0; }
    }
    // The reference's loop nest (seed_finder.hpp:134-160: for every read offset i, extend l while the bytes agree) as ONE loop whose iteration either
    // extends by one window or closes offset i: lanes of a wavefront that are in a long extension do not hold up the lanes that are between
    // extensions (with the nest, every offset costs the wavefront its longest extension).
    uint32_t i = 0;
    bool ext = false;
    uint64_t pos = 0, nxt = m ? ptr[qb + (uint64_t)(m - 1) * 64u] : 0;      // the pointer of offset i + 1 is asked for while offset i is worked on
    while (i < m) {
        if (!ext) {
            pos = nxt;
            if (i + 1 < m) nxt = ptr[qb + (uint64_t)(m - 2 - i) * 64u];
            ext = pos != prev_pos_plus_one;
        }
        if (ext) {
            bool more = false;
            if ((i + l) < m && (pos + l) < n) {
                bool bytes = !fast;
                if (fast) {
                    // one window: the bases both 2-bit words hold from here on (at most 32)
                    const uint32_t qi = (uint32_t)(i + l);
                    const uint64_t a = pos + l;
                    const uint32_t to = (uint32_t)a & 31u, po = qi & 31u;
                    uint64_t x = (F.text2[a >> 5] >> (2 * to)) ^ (F.pw[(qi >> 5) * F.pw_stride] >> (2 * po));
                    if (has_inv) x |= pat[mb + (uint64_t)(qi >> 5) * 64u] >> (2 * po);
                    uint32_t lim = 32u - (to > po ? to : po);
                    if (m - qi < lim) lim = m - qi;
                    if (n - a < (uint64_t)lim) lim = (uint32_t)(n - a);
                    uint32_t k = x ? (ctz64_(x) >> 1) : 32u;
                    if (k > lim) k = lim;
                    // codes say [a, a + k) agrees and a + k does not (or the window ends there).  A text byte outside A / C / G / T reads as code 0: inside
                    // [a, a + k) it may have "agreed" with an A, and at a + k it may be what a pattern byte outside A / C / G / T equals - either way the
                    // block is marked, and the bytes decide
                    const uint64_t b0 = a >> F.exc_sh, b1 = (a + k) >> F.exc_sh;
                    bytes = (((F.exc[b0 >> 5] >> (b0 & 31u)) | (F.exc[b1 >> 5] >> (b1 & 31u))) & 1u) != 0;
                    if (!bytes) {
                        n_cmp += k + (k < lim ? 1u : 0u);
                        l += k;
                        if (k) n_Ns = 0;                      // (the bytes that agreed are A / C / G / T)
                        more = k == lim;                      // no mismatch seen yet: next window (or the end of the pattern / the text: the test above)
                    }
                }
                if (bytes) {
                    while ((i + l) < m && (pos + l) < n) {
                        const uint32_t qi = (uint32_t)(i + l);
                        const uint8_t qc = pat_byte(pat, pb, m, qi, pc);
                        ++n_cmp;
                        if (qc != text_byte(text, pos + l, tc)) break;
                        if (qc == 'N') n_Ns++; else n_Ns = 0;
                        ++l;
                    }
                }
            }
            if (more) continue;
            ext = false;
        }
        if (l >= pl && n_Ns < l && l >= min_len) found(pos, l, i);
        pl = l;
        l = (l == 0 ? 0 : (l - 1));
        prev_pos_plus_one = pos + 1;
        ++i;
    }
    if (!EMIT) { cnt_m[task] = km; cnt_s[task] = ks; }
}

// ------------------------------------------------------------------------------------------------
// phi
// ------------------------------------------------------------------------------------------------
struct phi_tab_t {
    const moni_phi_t* recs;
    const uint32_t* dir;
};

MONI_HD void phi_step(const phi_tab_t P, const moni_consts_t& K, uint64_t i, uint64_t& out_pos, uint64_t& out_lcp) {
    const uint64_t slot = i >> K.phi_shift;
    uint32_t lo = P.dir[slot], hi = P.dir[slot + 1];
    while (lo < hi) {                                   // rank(i) = number of keys < i
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((P.recs[mid].w0 & MONI_POS_MASK) < i) lo = mid + 1; else hi = mid;
    }
    const uint32_t jr = lo ? lo - 1 : (uint32_t)K.r - 1;   // predecessor_rank_circular
    const moni_u64x2 v = *reinterpret_cast<const moni_u64x2*>(P.recs + jr);
    const uint64_t j = v.x & MONI_POS_MASK;
    const uint64_t delta = j < i ? i - j : i + 1;
    const uint64_t prev = v.y & MONI_POS_MASK;
    const uint64_t lcp = (v.x >> 40) | ((v.y >> 40) << 24);
    uint64_t p = prev + delta;
    if (p >= K.n) p -= K.n;                             // (prev_sample + delta) % n
    if (p >= K.n) p %= K.n;
    out_pos = p;
    out_lcp = lcp - delta + 1;                          // unsigned, as in the reference
}

// lceToRBounded(ra, a, b, len) of the `-n` form (ShapedSlp, an absent submodule; call sites seed_finder.hpp:354,367): the number of bytes the
// text suffixes at a and b share, counted up to len.  The text is 8-byte aligned and padded: eight bytes per compare.
MONI_HD uint64_t lce_bounded(const uint8_t* __restrict__ text, uint64_t a, uint64_t b, uint64_t len) {
    uint64_t l = 0;
    while (l < len) {
        const uint64_t x = load8_unaligned(text, a + l) ^ load8_unaligned(text, b + l);
        if (x) {
            uint64_t same = 0;
            while (!((x >> (8 * same)) & 0xFFu)) ++same;
            l += same;
            break;
        }
        l += 8;
    }
    return l < len ? l : len;
}

struct occ_args_t {
    phi_tab_t phi, phi_inv;
    const uint8_t* text;            // K.no_lcp: the LCP of a phi step is measured here
    const uint64_t* seq_starts;     // n_seq + 1
    const uint32_t* name_id;        // n_seq
    moni_mem_t* mems;
    const uint32_t* aux;
    const uint64_t* read_mem_off;
    uint64_t n_mems;
    uint64_t* occs;                 // FILL: final occurrence array
    uint64_t* tmp;                  // COUNT: first tmp_cap occurrences per seed; FILL: source for short lists
    uint64_t* lowers;               // lower suffix of every left half's parent (seed_finder.hpp:276)
    uint32_t tmp_cap;
    uint32_t filter_seeds;
    uint32_t n_seeds_thr;
    uint32_t pool_rows;             // per-name counter rows available
    uint32_t* pool;                 // pool_rows * n_seq counters
    uint32_t* pool_next;            // bump allocator
    uint32_t* error_flag;
    unsigned long long* counters;
};

struct walk_t {
    uint64_t total, filtered, kept;
    uint64_t last_kept;             // occs.back()
    uint32_t* names;                // per-name counters or nullptr
    uint64_t* out;                  // destination for kept occurrences or nullptr
    uint64_t out_cap;
    unsigned long long phi_steps;
};

MONI_HD uint32_t seq_of(const uint64_t* __restrict__ seq_starts, uint32_t n_seq, uint64_t pos) {
    // rank1(pos + 1) - 1  (seqidx.hpp:136-139): number of onsets <= pos, minus one
    uint32_t lo = 0, hi = n_seq + 1;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (seq_starts[mid] <= pos) lo = mid + 1; else hi = mid; }
    uint32_t k = lo ? lo - 1 : 0;
    return k < n_seq ? k : n_seq - 1;
}

// one occurrence (seed_finder.hpp:182-193 / 246-248): count it, apply the per-genome filter, keep or drop
MONI_HD void walk_push(walk_t& W, const occ_args_t& A, const moni_consts_t& K, uint64_t pos, bool first) {
    bool keep = true;
    if (W.names) {
        const uint32_t id = A.name_id[seq_of(A.seq_starts, K.n_seq, pos)];
        const uint32_t c = ++W.names[id];
        if (!first && A.filter_seeds && c > A.n_seeds_thr) keep = false;
    }
    W.total++;
    if (keep) {
        if (W.out && W.kept < W.out_cap) W.out[W.kept] = pos;
        W.kept++;
        W.last_kept = pos;
    } else W.filtered++;
}

// find_MEM_above / find_MEM_below (seed_finder.hpp:169-239 with the guards of 377-393)
template <bool ABOVE>
MONI_HD void walk_dir(walk_t& W, const occ_args_t& A, const moni_consts_t& K, uint64_t curr, uint64_t len) {
    while (true) {
        uint64_t nxt, lcp;
        if (ABOVE) {
            if (curr == K.first_run_sample) break;              // returns {last_run_sample, 0}: 0 >= len is false
            phi_step(A.phi, K, curr, nxt, lcp);
        } else {
            if (curr == K.last_run_sample) break;
            phi_step(A.phi_inv, K, curr, nxt, lcp);
        }
        W.phi_steps++;
        if (K.no_lcp) {                                         // seed_finder.hpp:346-370: no sampled LCP, a bounded LCE on the text when both suffixes are long enough
            lcp = 0;
            if (K.n_text - curr >= len && K.n_text - nxt >= len) lcp = lce_bounded(A.text, curr, nxt, len);
        }
        if (!(lcp >= len)) break;
        walk_push(W, A, K, nxt, false);
        curr = nxt;
    }
}

// [first_pos] + above(above_from) + below(below_from); full MEMs and right halves pass the same position three
// times (find_MEM_occs), left halves pass (upper, upper, lower) (seed_finder.hpp:283-290).
MONI_HD void walk_seed(walk_t& W, const occ_args_t& A, const moni_consts_t& K, uint64_t first_pos, uint64_t above_from,
                       uint64_t below_from, uint64_t len, uint64_t& upper, uint64_t& lower) {
    walk_push(W, A, K, first_pos, true);
    walk_dir<true>(W, A, K, above_from, len);
    upper = W.last_kept;
    walk_dir<false>(W, A, K, below_from, len);
    lower = W.last_kept;
}

// A walk can only lose occurrences to the per-genome filter when it sees more than n_seeds_thr of them in
// total; only then is it redone with per-name counters taken from the pool.
MONI_HD void run_seed(const occ_args_t& A, const moni_consts_t& K, uint64_t first_pos, uint64_t above_from, uint64_t below_from,
                      uint64_t len, uint64_t* out, uint64_t out_cap, walk_t& W, uint64_t& upper, uint64_t& lower) {
    W.total = W.filtered = W.kept = 0; W.last_kept = first_pos; W.names = nullptr; W.out = out; W.out_cap = out_cap;
    walk_seed(W, A, K, first_pos, above_from, below_from, len, upper, lower);
    if (A.filter_seeds && W.total > A.n_seeds_thr) {
        const uint32_t row = MONI_ATOMIC_INC_U32(A.pool_next);
        if (row >= A.pool_rows) { MONI_FLAG_SET(A.error_flag); return; }
        uint32_t* names = A.pool + (uint64_t)row * K.n_seq;
        for (uint32_t i = 0; i < K.n_seq; ++i) names[i] = 0;
        const unsigned long long ps = W.phi_steps;
        W.total = W.filtered = W.kept = 0; W.last_kept = first_pos; W.names = names;
        walk_seed(W, A, K, first_pos, above_from, below_from, len, upper, lower);
        W.phi_steps = ps;                                  // count the reference's walk once
    }
}

// occ_task<FILL>: one lane per final MEM slot.  A full MEM's lane also walks its left half (which starts
// from the MEM's upper/lower suffix); right halves have their own lane.
//   FILL = false: fills total_occ / num_filtered / occ_cnt (+ left-half pos, lower suffix) and keeps the
//                 first tmp_cap occurrences of every seed in tmp.
//   FILL = true : writes occurrences at occ_off (from tmp when the list fits, otherwise by walking again).
template <bool FILL>
MONI_HD void occ_task(const moni_consts_t& K, const occ_args_t& A, uint64_t g, unsigned long long& phi_steps) {
    const uint32_t ax = A.aux[g];
    if (ax == 0xFFFFFFFEu) return;                         // left halves ride with their parent
    const moni_mem_t M = A.mems[g];
    walk_t W; W.phi_steps = 0;
    uint64_t upper = 0, lower = 0;
    if (!FILL) {
        run_seed(A, K, M.pos, M.pos, M.pos, M.len, A.tmp + g * A.tmp_cap, A.tmp_cap, W, upper, lower);
        A.mems[g].total_occ = (uint32_t)W.total; A.mems[g].num_filtered = (uint32_t)W.filtered; A.mems[g].occ_cnt = (uint32_t)W.kept;
    } else if (M.occ_cnt <= A.tmp_cap) {
        for (uint32_t i = 0; i < M.occ_cnt; ++i) A.occs[M.occ_off + i] = A.tmp[g * A.tmp_cap + i];
    } else {
        run_seed(A, K, M.pos, M.pos, M.pos, M.len, A.occs + M.occ_off, M.occ_cnt, W, upper, lower);
    }
    if (ax < 0xFFFFFFFDu) {                                // this MEM has halves: its left half is walked here
        const uint64_t hb = A.read_mem_off[M.read] + ax;
        const moni_mem_t Bm = A.mems[hb];
        walk_t WB; WB.phi_steps = 0;
        uint64_t u2, l2;
        if (!FILL) {
            run_seed(A, K, upper, upper, lower, Bm.len, A.tmp + hb * A.tmp_cap, A.tmp_cap, WB, u2, l2);
            A.mems[hb].pos = upper;
            A.mems[hb].total_occ = (uint32_t)WB.total; A.mems[hb].num_filtered = (uint32_t)WB.filtered; A.mems[hb].occ_cnt = (uint32_t)WB.kept;
            A.lowers[hb] = lower;
            W.phi_steps += WB.phi_steps;
        } else if (Bm.occ_cnt <= A.tmp_cap) {
            for (uint32_t i = 0; i < Bm.occ_cnt; ++i) A.occs[Bm.occ_off + i] = A.tmp[hb * A.tmp_cap + i];
        } else {
            run_seed(A, K, Bm.pos, Bm.pos, A.lowers[hb], Bm.len, A.occs + Bm.occ_off, Bm.occ_cnt, WB, u2, l2);
        }
    }
    phi_steps += W.phi_steps;
}

// ------------------------------------------------------------------------------------------------
// genome_task: the MEM statistics of `-c` (calculate_MEM_stats, aligner_ksw2.hpp:1868-1902) need, for every seed, the largest and the smallest
// count its count_dict holds: how often it occurs on any one genome, filtered occurrences included (populate_dict counts before the per-genome
// cap drops an occurrence, seed_finder.hpp:331-343).  One lane per final seed slot repeats the slot's walk with a row of per-name counters of
// its own (rows: n slots x n_seq counters, zeroed by the caller) and leaves hi_lo[g] = largest | smallest << 32.
// ------------------------------------------------------------------------------------------------
MONI_HD void genome_task(const moni_consts_t& K, const occ_args_t& A, uint64_t g, uint64_t g0, uint32_t* __restrict__ rows, uint64_t* __restrict__ hi_lo) {
    const uint32_t ax = A.aux[g];
    const moni_mem_t M = A.mems[g];
    walk_t W; W.phi_steps = 0;
    W.total = W.filtered = W.kept = 0; W.last_kept = M.pos; W.names = rows + (g - g0) * K.n_seq; W.out = nullptr; W.out_cap = 0;
    uint64_t upper = 0, lower = 0;
    walk_seed(W, A, K, M.pos, M.pos, ax == 0xFFFFFFFEu ? A.lowers[g] : M.pos, M.len, upper, lower);      // a left half walks down from its parent's lower suffix
    uint32_t hi = 0, lo = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < K.n_seq; ++i) { const uint32_t v = W.names[i]; if (v) { hi = v > hi ? v : hi; lo = v < lo ? v : lo; } }
    hi_lo[g] = (uint64_t)hi | ((uint64_t)(hi ? lo : 0u) << 32);
}
