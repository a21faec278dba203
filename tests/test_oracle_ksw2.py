"""oracle/ksw2.hpp against (1) the reference's own worked example and (2) a naive Gotoh DP written
independently of the restatement (full matrices, -inf boundaries)."""
import json
import os

import numpy as np

from oracle import orc

NEG = -(1 << 29)


def nt(s):
    return np.array([int(c) for c in s], dtype=np.uint8)


def cig_str(c):
    return "".join("%d%s" % (x >> 4, "MID"[x & 15]) for x in c)


def test_reference_comment_example():
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ksw2_comment_example.json")))
    ext = orc.FLAG_EXTZ_ONLY | orc.FLAG_RIGHT
    left = orc.extz(nt(g["left"]["query"][::-1]), nt(g["left"]["target"][::-1]), ext)
    right = orc.extz(nt(g["right"]["query"]), nt(g["right"]["target"]), ext)
    assert left["reach_end"] == 1 and right["reach_end"] == 1
    assert left["mqe"] + right["mqe"] + 2 * g["mem_len"] == g["old_score"]
    glob = orc.extz(nt(g["global"]["query"]), nt(g["global"]["target"]), orc.FLAG_RIGHT)
    assert glob["score"] == g["new_score"]
    assert cig_str(glob["cigar"]) == g["global"]["cigar"]
    so = orc.extz(nt(g["global"]["query"]), nt(g["global"]["target"]), orc.FLAG_SCORE_ONLY)
    assert so["score"] == g["new_score"] and so["n_cigar"] == 0


def naive(q, t, qo=4, e=2):
    """H/E/F by the textbook recurrence; returns (H matrix)."""
    n, m = len(t), len(q)
    H = np.full((n + 1, m + 1), NEG, dtype=np.int64)
    E = np.full((n + 1, m + 1), NEG, dtype=np.int64)   # gap consuming target (deletion)
    F = np.full((n + 1, m + 1), NEG, dtype=np.int64)
    H[0, 0] = 0
    for i in range(1, n + 1):
        H[i, 0] = -(qo + i * e)
    for j in range(1, m + 1):
        H[0, j] = -(qo + j * e)
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            E[i, j] = max(H[i - 1, j] - qo, E[i - 1, j]) - e
            F[i, j] = max(H[i, j - 1] - qo, F[i, j - 1]) - e
            a, b = t[i - 1], q[j - 1]
            s = -e if (a == 4 or b == 4) else (2 if a == b else -4)
            H[i, j] = max(H[i - 1, j - 1] + s, E[i, j], F[i, j])
    return H[1:, 1:]


def path_score(q, t, cigar, i0, j0, qo=4, e=2):
    """score of the alignment a CIGAR spells, ending at (i0, j0), starting at (-1,-1)"""
    i = j = 0
    sc = 0
    for c in cigar:
        ln, op = int(c >> 4), int(c & 15)
        if op == 0:
            for _ in range(ln):
                a, b = t[i], q[j]
                sc += -e if (a == 4 or b == 4) else (2 if a == b else -4)
                i += 1; j += 1
        elif op == 2:
            sc -= qo + ln * e; i += ln
        else:
            sc -= qo + ln * e; j += ln
    assert (i - 1, j - 1) == (i0, j0), ((i, j), (i0, j0))
    return sc


def test_against_naive_gotoh():
    rng = np.random.default_rng(42)
    for it in range(300):
        m, n = int(rng.integers(1, 45)), int(rng.integers(1, 45))
        t = rng.integers(0, 4, size=n).astype(np.uint8)
        if rng.random() < 0.7:   # related sequences
            q = np.resize(t, m).copy()
            mut = rng.random(m) < 0.15
            q[mut] = rng.integers(0, 4, size=int(mut.sum()))
            if m > 4 and rng.random() < 0.5:
                k = int(rng.integers(1, m - 1)); q = np.concatenate([q[:k], q[k + 1:], q[:1]])
        else:
            q = rng.integers(0, 4, size=m).astype(np.uint8)
        if rng.random() < 0.2:
            q[int(rng.integers(0, m))] = 4
        if rng.random() < 0.2:
            t[int(rng.integers(0, n))] = 4
        H = naive(q, t)
        for flag in (orc.FLAG_SCORE_ONLY, orc.FLAG_RIGHT, orc.FLAG_EXTZ_ONLY | orc.FLAG_RIGHT, 0, orc.FLAG_EXTZ_ONLY):
            r = orc.extz(q, t, flag)
            assert r["score"] == H[n - 1, m - 1]
            col = H[:, m - 1]
            assert r["mqe"] == col.max() and r["mqe_t"] == int(np.argmax(col))      # first maximum
            row = H[n - 1, :]
            assert r["mte"] == row.max()
            mx = H.max()
            assert r["max"] == (mx if mx > 0 else 0)
            if mx > 0:
                assert H[r["max_t"], r["max_q"]] == mx
            if flag & orc.FLAG_SCORE_ONLY:
                assert r["n_cigar"] == 0
            elif not (flag & orc.FLAG_EXTZ_ONLY):
                assert path_score(q, t, r["cigar"], n - 1, m - 1) == r["score"]
            else:
                assert r["reach_end"] == 1    # end_bonus 400 dominates at these sizes
                assert path_score(q, t, r["cigar"], r["mqe_t"], m - 1) == r["mqe"]


def test_degenerate_inputs():
    q = np.array([0, 1, 2], dtype=np.uint8)
    r = orc.extz(q[:0], q, orc.FLAG_RIGHT)
    assert r["score"] == -0x40000000 and r["n_cigar"] == 0 and r["mqe_t"] == -1
    r = orc.extz(q, q[:0], orc.FLAG_SCORE_ONLY)
    assert r["mqe"] == -0x40000000 and r["max"] == 0
    r = orc.extz(q[:1], q[:1], orc.FLAG_RIGHT)
    assert r["score"] == 2 and cig_str(r["cigar"]) == "1M"
