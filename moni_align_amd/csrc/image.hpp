// Host-side conversion: semantic index arrays (moni_flat_index_t) -> device layout (layout.h).
// O(r) single pass per structure; runs once at load time, like seed_finder's constructor
// (include/aligner/seed_finder.hpp:64-124).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/moni_hip.h"
#include "layout.h"

struct HostImage {
    moni_consts_t K;
    moni_tables_t T;
    std::vector<moni_row_t> rows;      // r + 2
    std::vector<moni_frow_t> frows;    // r + 2
    std::vector<uint32_t> cr;          // (r + 1) * sigma
    std::vector<moni_rec_t> recs;
    std::vector<moni_phi_t> phi, phi_inv;   // r each
    std::vector<uint32_t> phi_dir, phi_inv_dir;
    std::vector<uint64_t> seq_starts;
    std::string err;

    static inline moni_row_t pack_row(uint64_t start, uint32_t head, uint64_t lfbase, uint64_t dest, uint64_t len = 0) {
        moni_row_t x;
        const uint64_t l = len > MONI_ROW_LEN_SAT ? MONI_ROW_LEN_SAT : len;
        x.w0 = start | ((uint64_t)head << 40) | ((dest >> 24) << 44) | (l << 52);
        x.w1 = lfbase | ((dest & 0xFFFFFFull) << 40);
        x.hot_cr[0] = x.hot_cr[1] = x.hot_cr[2] = x.hot_cr[3] = 0;
        return x;
    }

    // the per-run loops below are independent across runs: a few host threads share them (the image of a 46 M-run index is 10 GB)
    template <class Fn>
    static void par_for(uint64_t n, Fn fn) {
        unsigned T = std::thread::hardware_concurrency();
        if (const char* v = getenv("MONI_IMAGE_THREADS")) T = (unsigned)atoi(v);
        if (T > 16) T = 16;
        if (T < 2 || n < (1u << 16)) { fn((uint64_t)0, n); return; }
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; ++t) th.emplace_back(fn, n * t / T, n * (t + 1) / T);
        fn((uint64_t)0, n / T);
        for (auto& x : th) x.join();
    }

    int build(const moni_flat_index_t& f) {
        const uint64_t n = f.n, r = f.r;
        if (n >= (1ull << 40) || r >= (1ull << 32) - 2 || r == 0) { err = "index too large for the 40/32-bit layout"; return MONI_ERANGE; }
        memset(&K, 0, sizeof(K));
        memset(&T, 0, sizeof(T));
        K.n = n; K.r = r; K.n_text = n - 1; K.n_seq = (uint32_t)f.n_seq; K.no_lcp = f.slcp ? 0u : 1u;
        {   // phi / phi_inv (two sorts of r keys) are independent of everything else: two threads build them beside the rows
            uint32_t sh = 0;
            while ((n >> sh) > 2 * r + 1024 && sh < 30) ++sh;   // about <= 2 keys per directory slot on average
            K.phi_shift = sh;
        }
        int rc_phi[2] = {MONI_OK, MONI_OK}; std::string err_phi[2];
        std::thread phi_thread[2];
        phi_thread[0] = std::thread([&]() { rc_phi[0] = build_phi(f, f.ssa, false, phi, phi_dir, err_phi[0]); });
        phi_thread[1] = std::thread([&]() { rc_phi[1] = build_phi(f, f.esa, true, phi_inv, phi_inv_dir, err_phi[1]); });
        struct Join { std::thread* t; ~Join() { for (int i = 0; i < 2; ++i) if (t[i].joinable()) t[i].join(); } } join_phi{phi_thread};
        K.last_run_sample = (f.esa[r - 1] + 1) % n;
        K.first_run_sample = (f.ssa[0] + 1) % n;
        // alphabet
        uint64_t n_letter[256] = {0};
        uint64_t runs_letter[256] = {0};
        for (uint64_t k = 0; k < r; ++k) {
            if (f.starts[k + 1] <= f.starts[k]) { err = "run starts not increasing"; return MONI_ERANGE; }
            n_letter[f.heads[k]] += f.starts[k + 1] - f.starts[k];
            runs_letter[f.heads[k]]++;
        }
        if (f.starts[0] != 0 || f.starts[r] != n) { err = "run starts do not cover [0,n)"; return MONI_ERANGE; }
        uint32_t sigma = 0;
        int code_of[256];
        int byte_of[MONI_MAX_SIGMA];
        for (int b = 0; b < 256; ++b) {
            code_of[b] = -1;
            T.code[b] = MONI_CODE_ABSENT;
            if (n_letter[b]) {
                if (sigma >= MONI_MAX_SIGMA - 1) { err = "more than 15 distinct BWT symbols"; return MONI_ERANGE; }
                code_of[b] = (int)sigma; byte_of[sigma] = b; T.code[b] = (uint8_t)sigma; ++sigma;
            }
            T.compl_tab[b] = (uint8_t)b;
        }
        T.compl_tab['A'] = 'T'; T.compl_tab['C'] = 'G'; T.compl_tab['G'] = 'C'; T.compl_tab['T'] = 'A';
        T.compl_tab['a'] = 'T'; T.compl_tab['c'] = 'G'; T.compl_tab['g'] = 'C'; T.compl_tab['t'] = 'A';
        K.sigma = sigma;
        // F must be the cumulative letter counts (moni.hpp:253-282)
        {
            uint64_t acc = 0;
            for (int b = 0; b < 256; ++b) {
                if (f.F[b] != acc) { err = "F is not the cumulative symbol count"; return MONI_ERANGE; }
                acc += n_letter[b];
            }
        }
        // rows: start/head, then lfbase/dest in F order (one monotone pointer over the run starts)
        rows.assign(r + 2, pack_row(0, 0, 0, 0));
        std::vector<uint64_t> lfbase(r), dest(r);
        {
            std::vector<std::vector<uint32_t>> cruns(sigma);
            for (uint32_t c = 0; c < sigma; ++c) cruns[c].reserve(runs_letter[byte_of[c]]);
            for (uint64_t k = 0; k < r; ++k) cruns[code_of[f.heads[k]]].push_back((uint32_t)k);
            // recs
            uint32_t base = 0;
            for (uint32_t c = 0; c < sigma; ++c) { K.rec_base[c] = base; K.rec_cnt[c] = (uint32_t)cruns[c].size(); base += (uint32_t)cruns[c].size() + 1; }
            K.rec_base[sigma] = base;
            recs.assign(base, moni_rec_t{0, 0, 0, 0});
            uint64_t d = 0;   // run containing the current lf position
            for (uint32_t c = 0; c < sigma; ++c) {
                const int b = byte_of[c];
                uint64_t cum = 0;
                const size_t Rc = cruns[c].size();
                for (size_t j = 0; j <= Rc; ++j) {
                    const uint64_t lfpos = f.F[b] + cum;
                    while (d < r && f.starts[d + 1] <= lfpos) ++d;     // d == r  <=>  lfpos == n
                    moni_rec_t& R = recs[K.rec_base[c] + j];
                    uint64_t thr = 0, ssa = 0, esa_prev = 0;
                    if (j < Rc) {
                        const uint64_t k = cruns[c][j];
                        thr = f.thr[k]; ssa = f.ssa[k];
                        if ((j == 0) != (thr == 0)) { err = "threshold 0 must mark exactly the first run of a letter"; return MONI_ERANGE; }
                        if (j > 0) {
                            const uint64_t kp = cruns[c][j - 1];
                            // the device compares pos with thr_j only; that equals thr_bv::rank (thresholds_ds.hpp:494)
                            // iff every threshold lies in (end of previous c-run, start of this c-run]
                            if (!(thr > f.starts[kp + 1] - 1 && thr <= f.starts[k])) { err = "threshold outside its inter-run interval"; return MONI_ERANGE; }
                        }
                        lfbase[k] = lfpos; dest[k] = d;
                        cum += f.starts[k + 1] - f.starts[k];
                    }
                    if (j > 0) esa_prev = f.esa[cruns[c][j - 1]];
                    if ((thr | ssa | esa_prev | lfpos) >> 40) { err = "value exceeds 40 bits"; return MONI_ERANGE; }
                    R.w0 = thr | ((d >> 24) << 40);
                    R.w1 = ssa | ((d & 0xFFFFFFull) << 40);
                    R.w2 = esa_prev;
                    R.w3 = lfpos;
                }
            }
            // cr
            cr.assign((r + 1) * (uint64_t)sigma, 0);
            std::vector<uint32_t> seen(sigma, 0);
            for (uint64_t k = 0; k <= r; ++k) {
                for (uint32_t c = 0; c < sigma; ++c) cr[k * sigma + c] = seen[c];
                if (k < r) seen[code_of[f.heads[k]]]++;
            }
        }
        // hot symbols: the (up to) four codes with the most runs get their c-run count inside the row
        {
            for (uint32_t c = 0; c < MONI_MAX_SIGMA; ++c) K.hot_slot[c] = 0xFF;
            std::vector<std::pair<uint64_t, uint32_t>> by_runs;
            for (uint32_t c = 0; c < sigma; ++c) by_runs.push_back(std::make_pair(runs_letter[byte_of[c]], c));
            std::sort(by_runs.begin(), by_runs.end(), std::greater<std::pair<uint64_t, uint32_t>>());
            for (uint32_t s = 0; s < 4 && s < by_runs.size(); ++s) K.hot_slot[by_runs[s].second] = (uint8_t)s;
        }
        par_for(r + 1, [&](uint64_t k0, uint64_t k1) {
            for (uint64_t k = k0; k < k1; ++k) {
                rows[k] = k < r ? pack_row(f.starts[k], (uint32_t)code_of[f.heads[k]], lfbase[k], dest[k], f.starts[k + 1] - f.starts[k])
                                : pack_row(n, MONI_HEAD_NONE, 0, r, MONI_ROW_LEN_SAT);
                for (uint32_t c = 0; c < sigma; ++c) if (K.hot_slot[c] != 0xFF) rows[k].hot_cr[K.hot_slot[c]] = cr[k * sigma + c];
            }
        });
        rows[r + 1] = pack_row(MONI_POS_MASK, MONI_HEAD_NONE, 0, r, MONI_ROW_LEN_SAT);
        // fast rows
        {
            moni_frow_t z; memset(&z, 0, sizeof z);
            frows.assign(r + 2, z);
            int hot_code[4] = {-1, -1, -1, -1};
            for (uint32_t c = 0; c < sigma; ++c) if (K.hot_slot[c] != 0xFF) hot_code[K.hot_slot[c]] = (int)c;
            const bool four = hot_code[0] >= 0 && hot_code[1] >= 0 && hot_code[2] >= 0 && hot_code[3] >= 0;
            if (four) par_for(r, [&](uint64_t k0, uint64_t k1) {
            for (uint64_t k = k0; k < k1; ++k) {
                const uint32_t h = (uint32_t)code_of[f.heads[k]];
                const uint64_t len = f.starts[k + 1] - f.starts[k];
                const uint64_t doff = lfbase[k] - f.starts[dest[k]];
                bool ok = K.hot_slot[h] != 0xFF && len < MONI_ROW_LEN_SAT && doff < MONI_ROW_LEN_SAT;
                if (!ok) continue;
                const uint32_t hs = K.hot_slot[h];
                uint64_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                uint64_t ssa[3], esa[3];
                for (uint32_t sl = 0; sl < 3 && ok; ++sl) {
                    const uint32_t c = (uint32_t)hot_code[(hs + 1 + sl) & 3];
                    const uint32_t j = cr[k * sigma + c];
                    const moni_rec_t& R = recs[K.rec_base[c] + j];
                    const uint64_t thr = R.w0 & MONI_POS_MASK;
                    const uint64_t sd = ((R.w0 >> 40) << 24) | (R.w1 >> 40);
                    const uint64_t lfpos = R.w3;
                    const uint64_t sdoff = lfpos - f.starts[sd];
                    if (sdoff >= MONI_ROW_LEN_SAT) { ok = false; break; }
                    uint64_t thr_off;
                    if (j == 0) thr_off = 0;
                    else if (j == K.rec_cnt[c]) thr_off = len;
                    else thr_off = thr <= f.starts[k] ? 0 : (thr >= f.starts[k] + len ? len : thr - f.starts[k]);
                    ssa[sl] = R.w1 & MONI_POS_MASK; esa[sl] = R.w2;
                    w[1 + sl] = thr_off | (sdoff << 12) | (sd << 24) | ((ssa[sl] >> 32) << 56);
                }
                if (!ok) continue;
                w[0] = len | (doff << 12) | ((uint64_t)dest[k] << 24) | ((uint64_t)hs << 56) | (1ull << 58);
                w[4] = (ssa[0] & 0xFFFFFFFFull) | ((ssa[1] & 0xFFFFFFFFull) << 32);
                w[5] = (ssa[2] & 0xFFFFFFFFull) | ((esa[0] & 0xFFFFFFFFull) << 32);
                w[6] = (esa[1] & 0xFFFFFFFFull) | ((esa[2] & 0xFFFFFFFFull) << 32);
                w[7] = (esa[0] >> 32) | ((esa[1] >> 32) << 8) | ((esa[2] >> 32) << 16);
                memcpy(frows[k].w, w, sizeof w);
            }
            });
        }
        // absent bytes: LF(pos, b) = F[b]   (moni.hpp:583-588)
        for (int b = 0; b < 256; ++b) {
            T.abs_pos[b] = f.F[b];
            uint64_t k = (uint64_t)(std::upper_bound(f.starts, f.starts + r, f.F[b]) - f.starts) - 1;
            if (f.F[b] >= n) k = r;
            T.abs_run[b] = (uint32_t)k;
        }
        seq_starts.assign(f.seq_starts, f.seq_starts + f.n_seq + 1);
        for (int i = 0; i < 2; ++i) { phi_thread[i].join(); if (rc_phi[i]) { err = err_phi[i]; return rc_phi[i]; } }
        return MONI_OK;
    }

    // moni.hpp:186-251 (build_phi) + the lookups of moni_lcp.hpp:230-272 folded into one record per key
    int build_phi(const moni_flat_index_t& f, const uint64_t* smp, bool inverse, std::vector<moni_phi_t>& out,
                  std::vector<uint32_t>& dir, std::string& err) {
        const uint64_t n = f.n, r = f.r;
        std::vector<std::pair<uint64_t, uint64_t>> s(r);
        for (uint64_t i = 0; i < r; ++i) s[i] = std::make_pair(smp[i], i);
        std::sort(s.begin(), s.end());
        out.resize(r);
        for (uint64_t i = 0; i < r; ++i) {
            const uint64_t run = s[i].second;
            uint64_t prev = MONI_POS_MASK, lcp = 0;   // undefined entries (Phi of SA[0] / Phi_inv of SA[n-1]) are guarded by the caller
            if (!inverse) { if (run > 0) { prev = f.esa[run - 1]; lcp = f.slcp ? f.slcp[run] : 0; } }
            else { if (run + 1 < r) { prev = f.ssa[run + 1]; lcp = f.slcp ? f.slcp[run + 1] : 0; } }
            if ((lcp >> 40) || (s[i].first >> 40)) { err = "phi value exceeds 40 bits"; return MONI_ERANGE; }
            out[i].w0 = s[i].first | ((lcp & 0xFFFFFFull) << 40);
            out[i].w1 = prev | ((lcp >> 24) << 40);
            if (i > 0 && s[i].first == s[i - 1].first) { err = "duplicate SA sample"; return MONI_ERANGE; }
        }
        const uint32_t sh = K.phi_shift;
        const uint64_t slots = (n >> sh) + 2;
        dir.resize(slots + 1);
        uint64_t p = 0;
        for (uint64_t sl = 0; sl <= slots; ++sl) {
            const uint64_t lo = sl << sh;
            while (p < r && s[p].first < lo) ++p;
            dir[sl] = (uint32_t)p;
        }
        return MONI_OK;
    }

    uint64_t bytes() const {
        return rows.size() * sizeof(moni_row_t) + frows.size() * sizeof(moni_frow_t) + cr.size() * 4 + recs.size() * sizeof(moni_rec_t) +
               (phi.size() + phi_inv.size()) * sizeof(moni_phi_t) + (phi_dir.size() + phi_inv_dir.size()) * 4;
    }
};
