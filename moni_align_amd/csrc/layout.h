// Device layout of the r-index ("index image") shared by the host-side image builder and the
// HIP kernels.  All positions are < 2^40 (the reference stores 5-byte values: common.hpp:64-65
// THRBYTES/SSABYTES), run indices < 2^32.
//
// The reference reaches rank/select through sdsl bit-vectors and a wavelet tree
// (ms_rle_string.hpp:116-167): >= 8 dependent cache misses per LF step.  Here every query the
// hot loop makes is precomputed per run, so one step costs one 32-byte row read, plus one 4-byte
// and one 32-byte read on a threshold jump:
//
//   rows[k]   (k = 0..r+1)   start(40) head(4) dest(32) lfbase(40) len(12) hot_cr[4]   32 B / run
//       LF(start of run k) = lfbase, dest = run containing it; len = min(run length, 4095) so that
//       "is pos still inside this run" needs no second row; hot_cr[s] = number of c-runs before run k
//       for the four symbols with the most runs (A, C, G, T on DNA), the only ones a read can ask a
//       threshold jump for; rows[r] is the sentinel run [n, 2^40) that stands for
//       rle_string::run_of_position(n) == R.  One 32-byte aligned row = one 64-byte HBM request.
//   cr[k*sigma + c]          the same count for every symbol (used only for symbols without a hot slot)
//                            (ms_rle_string::run_and_head_rank .first)
//   recs[base[c] + j]        for the j-th c-run (j = 0..Rc):                     32 B / run
//       thr(40)   threshold of c-run j                      (thr_bv, thresholds_ds.hpp:478-497)
//       ssa(40)   samples_start of c-run j                  (moni.hpp:614)
//       esa(40)   samples_last of c-run j-1                 (moni.hpp:606)
//       lfpos(40) F[c] + #c before c-run j                  (run_and_head_rank .second)
//       dest(32)  run containing lfpos
//   phi records (sorted by key)  key(40) prev(40) lcp(40)                        16 B / run
//       one array for Phi_lcp (moni_lcp.hpp:253-272), one for Phi_inv_lcp (230-248), each with a
//       direct-mapped directory dir[pos >> sh] = lower_bound(keys, slot start).
#pragma once
#include <stdint.h>

#define MONI_MAX_SIGMA 16
#define MONI_CODE_ABSENT 0xFF
#define MONI_HEAD_NONE 15
#define MONI_POS_MASK ((1ull << 40) - 1)

#define MONI_ROW_LEN_SAT 4095u
struct alignas(32) moni_row_t {   // 32 bytes
    uint64_t w0;      // start | head << 40 | (dest >> 24) << 44 | min(len, 4095) << 52
    uint64_t w1;      // lfbase | (dest & 0xFFFFFF) << 40
    uint32_t hot_cr[4];
};

// Fast row (one 64-byte aligned record per run = ONE HBM request per LF step, threshold jumps included).  Everything is
// relative to run starts, so the LF loop carries (run, offset) and never needs an absolute BWT position:
//   w0      len(12) | doff(12) << 12 | dest(32) << 24 | head_slot(2) << 56 | ok(1) << 58
//           LF(run start) = offset doff inside run dest
//   w1..w3  slot s (the hot symbol (head_slot + 1 + s) & 3):  thr_off(12) | sdoff(12) << 12 | sdest(32) << 24 | ssa_hi(8) << 56
//           jump up  iff offset < thr_off   (thr_off folds "no c-run above/below" and the threshold position, clamped to the run)
//           down: sample = ssa, go to (sdest, sdoff);  up: sample = esa, go to one position before (sdest, sdoff)
//   w4..w7  the six 40-bit samples: ssa0|ssa1, ssa2|esa0, esa1|esa2 (low 32 bits), then the three esa high bytes
// ok = 0 (long runs >= 4095, offsets that do not fit 12 bits, heads or symbols outside the four hot ones, sentinels)
// sends the step down the general path over rows / cr / recs with absolute positions.
struct alignas(64) moni_frow_t { uint64_t w[8]; };
#define MONI_OFF_END 0xFFFFFFFFu

struct moni_rec_t {   // 32 bytes
    uint64_t w0;      // thr | (dest >> 24) << 40
    uint64_t w1;      // ssa | (dest & 0xFFFFFF) << 40
    uint64_t w2;      // esa of the previous c-run
    uint64_t w3;      // lfpos
};

struct moni_phi_t {   // 16 bytes
    uint64_t w0;      // key | (lcp & 0xFFFFFF) << 40
    uint64_t w1;      // prev | (lcp >> 24) << 40
};

// Small constant block copied to every kernel by value.
struct moni_consts_t {
    uint64_t n, r, n_text;
    uint64_t first_run_sample, last_run_sample;   // moni.hpp:331-333, r_index get_last_run_sample
    uint32_t sigma;
    uint32_t phi_shift;
    uint32_t n_seq;
    uint32_t no_lcp;                              // the index carries no LCP samples (<prefix>.thrbv.full.ms, `-n`): phi steps measure the LCP on the text (seed_finder.hpp:346-370)
    uint32_t rec_base[MONI_MAX_SIGMA + 1];
    uint32_t rec_cnt[MONI_MAX_SIGMA];             // Rc per code
    uint8_t hot_slot[MONI_MAX_SIGMA];             // code -> slot in moni_row_t::hot_cr, 0xFF if none
};

// 256-entry byte tables (kept in one device buffer, staged to LDS by the kernels)
struct moni_tables_t {
    uint8_t code[256];        // byte -> code, MONI_CODE_ABSENT if the byte does not occur in the BWT
    uint8_t compl_tab[256];   // kpbseq.h:120-137
    uint32_t abs_run[256];    // run containing F[b]   (n_c == 0 branch, moni.hpp:583-588)
    uint64_t abs_pos[256];    // F[b]
};
