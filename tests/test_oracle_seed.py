"""The CPU oracle (oracle/seed.hpp) against brute force that does not depend on the restatement:
matching statistics by direct search, phi by a naive suffix array, occurrences by exhaustive scan."""
import numpy as np
import pytest

from oracle import orc


def brute_ms_len(text: bytes, p: bytes, i: int) -> int:
    lo, hi = 0, len(p) - i
    while lo < hi:   # longest prefix of p[i:] that occurs in text (monotone)
        mid = (lo + hi + 1) // 2
        if text.find(p[i:i + mid]) >= 0:
            lo = mid
        else:
            hi = mid - 1
    return lo


def all_occ(text: bytes, s: bytes):
    out, k = [], text.find(s)
    while k >= 0:
        out.append(k)
        k = text.find(s, k + 1)
    return out


@pytest.fixture(scope="module")
def oidx(small_case):
    return orc.OracleIndex(small_case.path)


def test_ms_pointers_are_true_matching_statistics(small_case, oidx):
    text = small_case.text
    reads = small_case.synth.make_reads(small_case.pg, 40, 100, seed=7, sub_rate=0.03, indel_rate=0.002)
    reads[3, 10] = ord("N")
    reads[4, 50:53] = ord("n")
    for r in reads:
        p = r.tobytes()
        ptr = oidx.ms_query(p)
        for i in range(len(p)):
            l = brute_ms_len(text, p, i)
            if l > 0:
                assert text[int(ptr[i]):int(ptr[i]) + l] == p[i:i + l], (i, l)


def test_phi_matches_naive_suffix_array(small_case, oidx):
    tb = small_case.text + b"\x00"
    n = len(tb)
    sa = sorted(range(n), key=lambda i: tb[i:])
    lcp = [0] * n
    for j in range(1, n):
        a, b = sa[j - 1], sa[j]
        l = 0
        while tb[a + l] == tb[b + l]:
            l += 1
        lcp[j] = l
    for j in range(1, n, 7):
        assert oidx.phi_lcp(sa[j]) == (sa[j - 1], lcp[j])
    for j in range(0, n - 1, 7):
        assert oidx.phi_lcp(sa[j], inverse=True) == (sa[j + 1], lcp[j + 1])


def test_mems_and_occurrences(small_case, oidx):
    text = small_case.text
    L = 100
    reads = small_case.synth.make_reads(small_case.pg, 30, L, seed=11, sub_rate=0.02)
    offs = np.arange(0, (len(reads) + 1) * L, L, dtype=np.uint64)
    res = oidx.seed_batch(reads.reshape(-1), offs, min_len=25, filter_seeds=False, n_seeds_thr=1000)
    assert len(res["pos"]) > 0
    rc = small_case.synth.revcomp(reads)
    n_full = 0
    for k in range(len(res["pos"])):
        rd = int(res["read"][k])
        seq = (rc if (int(res["mate"][k]) & 2) else reads)[rd].tobytes()
        i, l, pos = int(res["idx"][k]), int(res["len"][k]), int(res["pos"][k])
        s = seq[i:i + l]
        occ = [int(x) for x in res["occs"][int(res["occ_off"][k]):int(res["occ_off"][k]) + int(res["occ_cnt"][k])]]
        assert occ[0] == pos and text[pos:pos + l] == s
        for o in occ:
            assert text[o:o + l] == s
        assert len(set(occ)) == len(occ)
        truth = set(all_occ(text, s))
        assert set(occ) <= truth
        # a MEM that is not a left half enumerates every occurrence (left halves skip the parent's interior)
        if set(occ) == truth:
            n_full += 1
        assert int(res["total_occ"][k]) == len(occ)
    assert n_full >= len(res["pos"]) // 2


def test_filter_counts(small_case, oidx):
    L = 100
    reads = small_case.synth.make_reads(small_case.pg, 10, L, seed=3, sub_rate=0.0, indel_rate=0.0)
    offs = np.arange(0, (len(reads) + 1) * L, L, dtype=np.uint64)
    a = oidx.seed_batch(reads.reshape(-1), offs, 25, False, 0)
    b = oidx.seed_batch(reads.reshape(-1), offs, 25, True, 0)
    assert np.array_equal(a["total_occ"], b["total_occ"])
    assert (b["num_filtered"] + b["occ_cnt"] == b["total_occ"]).all()
    assert (a["num_filtered"] == 0).all()
