// Lift-over of positions and CIGARs from a haplotype to the reference contig it was built from: liftidx::lift / lift_cigar
// (include/aligner/liftidx.hpp:89-95,159-164) over levioSAM's lift::Lift (absent submodule), used at
// include/aligner/aligner_ksw2.hpp:565-576 (check_left_MEM), :444 (score.lft) and :3133-3175 (final record).
//
// levioSAM keeps three Elias-Fano bit-vectors over the columns of the haplotype-vs-reference alignment (ins: the column has
// no reference base, del: the column has no haplotype base) and answers lift_pos(p) = ins.rank0(del.select0(p + 1)) with two
// rank/select descents.  Here a lift is the list of its maximal column runs with constant (ins, del) flags, each run
// carrying the three coordinates of its first column (alignment column, haplotype position, reference position): one
// binary search over a few thousand 16-byte records answers lift_pos, and lift_cigar walks runs instead of single columns.
// Compiled for the device (align kernels) and the host (host pipeline, record formatting).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LIFT_HD __host__ __device__ __forceinline__
#else
#define LIFT_HD inline
#endif

#define MONI_LIFT_INS 1u
#define MONI_LIFT_DEL 2u

struct moni_lift_run_t { uint32_t col, hap, ref, flags; };      // first column of the run and the haplotype / reference positions there
struct moni_lift_seq_t {
    uint64_t second;          // liftidx::lifts[i].second: where the target contig starts in the concatenation
    uint32_t run_off, n_runs; // runs[run_off .. run_off + n_runs); the last one is a sentinel at column = number of columns
                              // (flags 0: positions past the end continue as matches; the reference reads past its bit-vectors there)
    uint64_t start, end;      // the sequence's onset in the text and the next sequence's (seqidx::starts)
};
// Position directory: for every 2^shift-th text position the sequence it lies in (low 32 bits) and the lift run that holds its
// haplotype position (high 32 bits, index into the run array): seqidx::index and liftidx::lift of any position cost one directory
// entry, one moni_lift_seq_t and one or two runs instead of two binary searches of dependent loads.
#define MONI_PDIR_SHIFT 12

// the run that holds haplotype position p (a run without the del flag), and the column of p
LIFT_HD uint32_t lift_find(const moni_lift_run_t* __restrict runs, uint32_t n_runs, uint64_t p, uint64_t& x) {
    uint32_t lo = 0, hi = n_runs;                 // last run with hap <= p (runs[0].hap == 0)
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)runs[mid].hap <= p) lo = mid; else hi = mid; }
    x = (uint64_t)runs[lo].col + (p - (uint64_t)runs[lo].hap);
    return lo;
}

// lift_find when a run at or before p's run is already known (the position directory gives one): a short walk forward
LIFT_HD uint32_t lift_find_from(const moni_lift_run_t* __restrict runs, uint32_t n_runs, uint32_t hint, uint64_t p, uint64_t& x) {
    if (hint >= n_runs || (uint64_t)runs[hint].hap > p) return lift_find(runs, n_runs, p, x);
    uint32_t k = hint;
    for (int step = 0; step < 32 && k + 1 < n_runs; ++step) { if ((uint64_t)runs[k + 1].hap > p) { x = (uint64_t)runs[k].col + (p - (uint64_t)runs[k].hap); return k; } ++k; }
    if (k + 1 >= n_runs) { x = (uint64_t)runs[k].col + (p - (uint64_t)runs[k].hap); return k; }
    uint32_t lo = k, hi = n_runs;               // far from the hint: finish with the binary search
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)runs[mid].hap <= p) lo = mid; else hi = mid; }
    x = (uint64_t)runs[lo].col + (p - (uint64_t)runs[lo].hap);
    return lo;
}

// lift::Lift::lift_pos: ins.rank0(del.select0(p + 1))
LIFT_HD uint64_t lift_pos(const moni_lift_run_t* __restrict runs, uint32_t n_runs, uint64_t p) {
    uint64_t x;
    const uint32_t k = lift_find(runs, n_runs, p, x);
    return (uint64_t)runs[k].ref + ((runs[k].flags & MONI_LIFT_INS) ? 0ull : x - (uint64_t)runs[k].col);
}

// lift::Lift::lift_cigar for an alignment whose first reference-consuming column is haplotype position p: the unit-operation
// walk of levioSAM (a deleted column emits D and consumes nothing of the CIGAR; I stays; M becomes I on an inserted column;
// D / N vanish on an inserted column; equal neighbours merge, zero-length operations disappear), taken run by run.
// Returns the number of operations written, or -1 when they do not fit cap.
LIFT_HD int lift_cigar(const moni_lift_run_t* __restrict runs, uint32_t n_runs, uint64_t p, const uint32_t* __restrict cig, uint32_t n_cig,
                       uint32_t* __restrict out, uint32_t cap, uint32_t hint = 0xFFFFFFFFu, uint64_t* lifted_pos = nullptr) {
    uint64_t x;
    uint32_t k = lift_find_from(runs, n_runs, hint, p, x);
    if (lifted_pos) *lifted_pos = (uint64_t)runs[k].ref + ((runs[k].flags & MONI_LIFT_INS) ? 0ull : x - (uint64_t)runs[k].col);      // lift_pos(p)
    int n = 0;
    uint32_t cur_op = 0xFu; uint64_t cur_len = 0;
    bool ovf = false;
    auto push = [&](uint32_t op, uint64_t len) {
        if (len == 0) return;
        if (op == cur_op) { cur_len += len; return; }
        if (cur_op != 0xFu) { if ((uint32_t)n < cap) out[n++] = (uint32_t)(cur_len << 4) | cur_op; else ovf = true; }
        cur_op = op; cur_len = len;
    };
    for (uint32_t ci = 0; ci < n_cig; ++ci) {
        const uint32_t op = cig[ci] & 0xfu;
        uint64_t rem = cig[ci] >> 4;
        while (rem > 0) {
            const bool last = k + 1 >= n_runs;
            const uint64_t end = last ? ~0ull : (uint64_t)runs[k + 1].col;
            if (x >= end) { ++k; continue; }
            const uint64_t avail = end - x;
            const uint32_t fl = last ? 0u : runs[k].flags;
            if (fl & MONI_LIFT_DEL) { push(2u, avail); x += avail; continue; }
            if (op == 1u || op == 4u) { push(op, rem); rem = 0; continue; }
            const uint64_t t = rem < avail ? rem : avail;
            if (op == 0u || op == 7u || op == 8u) push((fl & MONI_LIFT_INS) ? 1u : 0u, t);
            else if (op == 2u || op == 3u) { if (!(fl & MONI_LIFT_INS)) push(op, t); }
            else { rem = 0; continue; }                     // H / P / B: skipped
            x += t; rem -= t;
        }
    }
    if (cur_op != 0xFu) { if ((uint32_t)n < cap) out[n++] = (uint32_t)(cur_len << 4) | cur_op; else ovf = true; }
    return ovf ? -1 : n;
}
