// ORACLE — TEST INFRASTRUCTURE ONLY (see flat_index.hpp header).  PARITY UNPINNED.
//
// ksw2.hpp: scalar restatement of lh3/ksw2 `ksw_extz2_sse` (ksw2_extz2_sse.c) and `ksw_backtrack`
// (ksw2.h).  thirdparty/ksw2 is an empty, unpinned submodule (GIT_TAG master,
// thirdparty/CMakeLists.txt:415-419), so this follows the published algorithm (Suzuki–Kasahara
// difference recurrence == plain affine-gap DP in exact integers) as the reference calls it:
//   include/aligner/aligner_ksw2.hpp:2812,2844,2965,2988,3015  (m=5, mat 2/-4, q=4, e=2, w=-1, zdrop=-1,
//   end_bonus=400, flags SCORE_ONLY | EXTZ_ONLY|RIGHT | RIGHT).
// Only the full-matrix case (w < 0 or w >= max(qlen,tlen)) and the exact-max path are restated; the
// reference never uses a band, GENERIC_SC or APPROX_MAX on this path.
//
//   H(i,j) = max{H(i-1,j-1)+s(i,j), E(i,j), F(i,j)}        i: target, j: query
//   E(i+1,j) = max{H(i,j)-q, E(i,j)} - e ;  F(i,j+1) = max{H(i,j)-q, F(i,j)} - e
//   H(-1,-1)=0, H(i,-1)=-(q+(i+1)e), H(-1,j)=-(q+(j+1)e), E(0,j)=H(-1,j)-q-e, F(i,0)=H(i,-1)-q-e
//   s: equal -> mat[0]; different -> mat[1]; either code == m-1 -> (mat[m*m-1]==0 ? -e : mat[m*m-1])
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace oracle {

#define KSW_NEG_INF -0x40000000
#define KSW_EZ_SCORE_ONLY 0x01
#define KSW_EZ_RIGHT 0x02
#define KSW_EZ_GENERIC_SC 0x04
#define KSW_EZ_APPROX_MAX 0x08
#define KSW_EZ_APPROX_DROP 0x10
#define KSW_EZ_EXTZ_ONLY 0x40
#define KSW_EZ_REV_CIGAR 0x80

typedef struct {
    uint32_t max : 31, zdropped : 1;
    int max_q, max_t;
    int mqe, mqe_t;
    int mte, mte_q;
    int score;
    int m_cigar, n_cigar;
    int reach_end;
    uint32_t* cigar;
} ksw_extz_t;

static inline void ksw_reset_extz(ksw_extz_t* ez) {
    ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
    ez->max = 0, ez->score = ez->mqe = ez->mte = KSW_NEG_INF;
    ez->n_cigar = 0, ez->zdropped = 0, ez->reach_end = 0;
}

static inline uint32_t* ksw_push_cigar(int* n_cigar, int* m_cigar, uint32_t* cigar, uint32_t op, int len) {
    if (*n_cigar == 0 || op != (cigar[(*n_cigar) - 1] & 0xf)) {
        if (*n_cigar == *m_cigar) {
            *m_cigar = *m_cigar ? (*m_cigar) << 1 : 4;
            cigar = (uint32_t*)realloc(cigar, (size_t)(*m_cigar) << 2);
        }
        cigar[(*n_cigar)++] = len << 4 | op;
    } else cigar[(*n_cigar) - 1] += len << 4;
    return cigar;
}

static inline int ksw_apply_zdrop_rot(ksw_extz_t* ez, int32_t H, int r, int t, int zdrop, int8_t e) {
    if (H > (int32_t)ez->max) {
        ez->max = H, ez->max_t = t, ez->max_q = r - t;
    } else if (t >= ez->max_t && r - t >= ez->max_q) {
        int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l;
        l = tl > ql ? tl - ql : ql - tl;
        if (zdrop >= 0 && (int32_t)ez->max - H > zdrop + l * e) {
            ez->zdropped = 1;
            return 1;
        }
    }
    return 0;
}

// p: direction bytes, row-major by anti-diagonal r with `n_col` bytes per diagonal, cell (r, i) at p[r*n_col + i - off[r]]
static inline void ksw_backtrack_rot(const uint8_t* p, const int* off, size_t n_col, int i0, int j0,
                                     int* m_cigar_, int* n_cigar_, uint32_t** cigar_) {
    int n_cigar = 0, m_cigar = *m_cigar_, i = i0, j = j0, r, state = 0;
    uint32_t* cigar = *cigar_, tmp;
    while (i >= 0 && j >= 0) {
        r = i + j;
        tmp = p[(size_t)r * n_col + i - off[r]];
        if (state == 0) state = tmp & 7;
        else if (!(tmp >> (state + 2) & 1)) state = 0;
        if (state == 0) state = tmp & 7;
        if (state == 0) cigar = ksw_push_cigar(&n_cigar, &m_cigar, cigar, 0, 1), --i, --j;
        else if (state == 1 || state == 3) cigar = ksw_push_cigar(&n_cigar, &m_cigar, cigar, 2, 1), --i;
        else cigar = ksw_push_cigar(&n_cigar, &m_cigar, cigar, 1, 1), --j;
    }
    if (i >= 0) cigar = ksw_push_cigar(&n_cigar, &m_cigar, cigar, 2, i + 1);
    if (j >= 0) cigar = ksw_push_cigar(&n_cigar, &m_cigar, cigar, 1, j + 1);
    for (i = 0; i < n_cigar >> 1; ++i)
        tmp = cigar[i], cigar[i] = cigar[n_cigar - 1 - i], cigar[n_cigar - 1 - i] = tmp;
    *m_cigar_ = m_cigar, *n_cigar_ = n_cigar, *cigar_ = cigar;
}

struct ksw_counters { uint64_t cells = 0, calls = 0; };

inline void ksw_extz2_restated(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int8_t m,
                               const int8_t* mat, int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag,
                               ksw_extz_t* ez, ksw_counters* kc = nullptr) {
    const int qe = q + e;
    const int with_cigar = !(flag & KSW_EZ_SCORE_ONLY);
    ksw_reset_extz(ez);
    if (m <= 0 || qlen <= 0 || tlen <= 0) return;
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    int max_sc = mat[0], min_sc = mat[1];
    for (int t = 1; t < m * m; ++t) {
        max_sc = max_sc > mat[t] ? max_sc : mat[t];
        min_sc = min_sc < mat[t] ? min_sc : mat[t];
    }
    if (-min_sc > 2 * (q + e)) return;
    if (w < (tlen > qlen ? tlen : qlen)) { abort(); }   // banded case not restated (never used by the reference)
    if (kc) { kc->cells += (uint64_t)qlen * tlen; kc->calls++; }
    const int sc_mch = mat[0], sc_mis = mat[1], sc_N = mat[m * m - 1] == 0 ? -e : mat[m * m - 1];

    // per-target-row state carried across anti-diagonals: H(i, j-1) , E(i,j) (for cell below), F handled per diagonal
    // We keep full rows of the previous two diagonals in (i)-indexed arrays.
    std::vector<int32_t> Hprev(tlen + 1), Hprev2(tlen + 1), Hcur(tlen + 1);   // H on diagonals r-1, r-2, r
    std::vector<int32_t> Eprev(tlen + 1), Ecur(tlen + 1);                       // E(i,j) stored at index i
    std::vector<int32_t> Fprev(tlen + 1), Fcur(tlen + 1);                       // F(i,j) stored at index i
    std::vector<int32_t> Hlin(tlen);                                              // ksw2's H[t]: last computed H in row t
    const size_t n_col = (size_t)tlen;
    std::vector<uint8_t> p;
    std::vector<int> off;
    if (with_cigar) { p.assign((size_t)(qlen + tlen - 1) * n_col, 0); off.assign(qlen + tlen - 1, 0); }

    auto Hb = [&](int i, int j) -> int32_t {   // boundary / lookup helper for cells outside the matrix
        if (i == -1 && j == -1) return 0;
        if (i == -1) return -(q + (j + 1) * e);
        return -(q + (i + 1) * e);             // j == -1
    };

    for (int r = 0; r < qlen + tlen - 1; ++r) {
        int st0 = 0, en0 = tlen - 1;
        if (st0 < r - qlen + 1) st0 = r - qlen + 1;
        if (en0 > r) en0 = r;
        const int en = (en0 + 16) / 16 * 16 - 1;   // ksw2 rounds the band end up to a 16-multiple (only used for mte_q)
        for (int i = st0; i <= en0; ++i) {
            const int j = r - i;
            // neighbours
            int32_t h_diag = (i > 0 && j > 0) ? Hprev2[i - 1] : Hb(i - 1, j - 1);
            int32_t h_up = (i > 0) ? Hprev[i - 1] : Hb(-1, j);       // H(i-1, j)
            int32_t h_left = (j > 0) ? Hprev[i] : Hb(i, -1);         // H(i, j-1)
            int32_t E, F;
            if (i > 0) { int32_t o = h_up - q, x = Eprev[i - 1]; E = (o > x ? o : x) - e; }
            else E = h_up - q - e;
            if (j > 0) { int32_t o = h_left - q, x = Fprev[i]; F = (o > x ? o : x) - e; }
            else F = h_left - q - e;
            const uint8_t tc = target[i], qc = query[j];
            int s = (tc == (uint8_t)(m - 1) || qc == (uint8_t)(m - 1)) ? sc_N : (tc == qc ? sc_mch : sc_mis);
            int32_t z = h_diag + s;
            uint8_t d;
            if (!(flag & KSW_EZ_RIGHT)) {
                d = E > z ? 1 : 0;
                z = z > E ? z : E;
                d = F > z ? 2 : d;
                z = z > F ? z : F;
            } else {
                d = z > E ? 0 : 1;
                z = z > E ? z : E;
                d = z > F ? d : 2;
                z = z > F ? z : F;
            }
            // continuation flags: is E(i+1,j) / F(i,j+1) an extension of E(i,j) / F(i,j)?
            if (!(flag & KSW_EZ_RIGHT)) {
                if (E > z - q) d |= 0x08;
                if (F > z - q) d |= 0x10;
            } else {
                if (E >= z - q) d |= 0x08;
                if (F >= z - q) d |= 0x10;
            }
            Hcur[i] = z; Ecur[i] = E; Fcur[i] = F;
            if (with_cigar) p[(size_t)r * n_col + i] = d;
        }
        // ---- exact max with ksw2's lane order (ksw2_extz2_sse.c "find the exact max") ----
        int32_t max_H, max_t;
        {
            for (int i = st0; i <= en0; ++i) Hlin[i] = Hcur[i];
            if (r > 0) {
                int en1 = st0 + (en0 - st0) / 4 * 4, t, i;
                int32_t HH[4], tt[4];
                max_H = Hlin[en0]; max_t = en0;
                for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
                for (t = st0; t < en1; t += 4)
                    for (i = 0; i < 4; ++i)
                        if (Hlin[t + i] > HH[i]) HH[i] = Hlin[t + i], tt[i] = t;
                for (i = 0; i < 4; ++i)
                    if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
                for (; t < en0; ++t)
                    if (Hlin[t] > max_H) max_H = Hlin[t], max_t = t;
            } else max_H = Hlin[0], max_t = 0;
        }
        if (en0 == tlen - 1 && Hlin[en0] > ez->mte) ez->mte = Hlin[en0], ez->mte_q = r - en;
        if (r - st0 == qlen - 1 && Hlin[st0] > ez->mqe) ez->mqe = Hlin[st0], ez->mqe_t = st0;
        if (ksw_apply_zdrop_rot(ez, max_H, r, max_t, zdrop, e)) break;
        if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = Hlin[tlen - 1];
        Hprev2.swap(Hprev); Hprev.swap(Hcur);
        Eprev.swap(Ecur); Fprev.swap(Fcur);
    }
    if (with_cigar) {
        if (!ez->zdropped && !(flag & KSW_EZ_EXTZ_ONLY)) {
            ksw_backtrack_rot(p.data(), off.data(), n_col, tlen - 1, qlen - 1, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
        } else if (!ez->zdropped && (flag & KSW_EZ_EXTZ_ONLY) && ez->mqe + end_bonus > (int)ez->max) {
            ez->reach_end = 1;
            ksw_backtrack_rot(p.data(), off.data(), n_col, ez->mqe_t, qlen - 1, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
        } else if (ez->max_t >= 0 && ez->max_q >= 0) {
            ksw_backtrack_rot(p.data(), off.data(), n_col, ez->max_t, ez->max_q, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
        }
    }
}

}  // namespace oracle
