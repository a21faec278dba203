"""index_build (torch prefix doubling) against a naive suffix sort on a small text."""
import collections

import numpy as np


def test_flat_index_matches_naive(small_case):
    fi = small_case.fi
    tb = small_case.text + b"\x00"
    n = len(tb)
    sa = sorted(range(n), key=lambda i: tb[i:])
    bwt = [tb[(i - 1) % n] for i in sa]
    bwt = [1 if b <= 1 else b for b in bwt]
    lcp = [0] * n
    for j in range(1, n):
        a, b = sa[j - 1], sa[j]
        l = 0
        while a + l < n and b + l < n and tb[a + l] == tb[b + l]:
            l += 1
        lcp[j] = l
    starts = [j for j in range(n) if j == 0 or bwt[j] != bwt[j - 1]]
    ends = starts[1:] + [n]
    assert fi.n == n and fi.r == len(starts)
    assert starts == list(fi.starts[:-1]) and int(fi.starts[-1]) == n
    assert [bwt[s] for s in starts] == list(fi.heads)
    assert [(sa[s] - 1) % n for s in starts] == list(fi.ssa)
    assert [(sa[e - 1] - 1) % n for e in ends] == list(fi.esa)
    assert [lcp[s] for s in starts] == list(fi.slcp)
    last_end = {}
    thr = []
    for k, s in enumerate(starts):
        c = bwt[s]
        if c not in last_end:
            thr.append(0)
        else:
            best = None
            for p in range(last_end[c] + 1, s + 1):
                if best is None or lcp[p] < lcp[best]:
                    best = p
            thr.append(best)
        last_end[c] = ends[k] - 1
    assert thr == list(fi.thr)
    cnt = collections.Counter(bwt)
    F = [0] * 256
    for c in range(1, 256):
        F[c] = F[c - 1] + cnt.get(c - 1, 0)
    assert F == list(fi.F)


def test_save_load_roundtrip(small_case, tmp_path):
    from moni_align_amd.index_build import FlatIndex
    g = FlatIndex.load(small_case.path)
    fi = small_case.fi
    for k in ("F", "heads", "starts", "ssa", "esa", "thr", "slcp", "text", "seq_starts"):
        assert np.array_equal(getattr(g, k), getattr(fi, k)), k
    assert g.names == fi.names and g.n == fi.n and g.r == fi.r and g.w == fi.w


def test_text_layout(small_case):
    # every sequence is followed by w bytes <= 5, the last by w + (w-1)  (test/src/ldx_slp_test.cpp:101-137)
    pg, fi = small_case.pg, small_case.fi
    t = fi.text
    acc = 0
    for i, s in enumerate(pg.seqs):
        assert (t[acc:acc + len(s)] > 5).all()
        acc += len(s)
        last = pg.w + (pg.w - 1 if i == len(pg.seqs) - 1 else 0)
        assert (t[acc:acc + last] <= 5).all()
        assert int(fi.seq_starts[i + 1]) == acc + pg.w
        acc += last
    assert acc == len(t)


def test_wide_doubling_and_32_bit_rank_levels_equal_the_default_path():
    """the paths a text of 2^31 positions or more takes - a doubling level as two stable sorts (the packed key rank * (n + 1) + rank' would
    overflow 64 bits beyond n = 3.03e9), level ranks kept as the low 32 bits with explicit wrap-around - forced on a small text: same suffix
    array, same LCP array; and the wrap-around of ranks at and beyond 2^31 keeps distinct ranks distinct"""
    import torch
    from moni_align_amd import index_build as ib
    rng = np.random.default_rng(3)
    unit = rng.integers(1, 5, size=400)
    codes = np.concatenate([unit, unit[:250], rng.integers(1, 5, size=300), unit, [0]]).astype(np.uint8)      # repeats: many doubling levels
    c = torch.from_numpy(codes)
    sa0, key0, k0, lv0 = ib.suffix_array(c, 3)
    sa1, key1, k1, lv1 = ib.suffix_array(c, 3, _force_wide=True)
    assert torch.equal(sa0, sa1) and k0 == k1 and len(lv0) == len(lv1) and len(lv0) >= 4
    assert all(torch.equal(a[1], b[1]) for a, b in zip(lv0, lv1))
    assert torch.equal(ib.lcp_from_levels(sa0, key0, k0, 3, lv0), ib.lcp_from_levels(sa1, key1, k1, 3, lv1))
    # n >= 2^32 (a GRCh38-scale pangenome): the levels keep the whole 64-bit rank - same arrays
    sa2, key2, k2, lv2 = ib.suffix_array(c, 3, _force_wide=True, rank64=True)
    assert torch.equal(sa0, sa2) and all(b[1].dtype == torch.int64 and torch.equal(a[1].to(torch.int64), b[1]) for a, b in zip(lv0, lv2))
    assert torch.equal(ib.lcp_from_levels(sa0, key0, k0, 3, lv0), ib.lcp_from_levels(sa2, key2, k2, 3, lv2))
    r = torch.tensor([0, 5, (1 << 31) - 1, 1 << 31, (1 << 31) + 7, (1 << 32) - 1], dtype=torch.int64)
    w = ib._rank32(r)
    assert w.dtype == torch.int32 and len(set(w.tolist())) == len(r) and w[:3].tolist() == [0, 5, (1 << 31) - 1]


def test_rank64_build_gives_the_same_index(small_case, tmp_path, monkeypatch):
    """MONI_BUILD_RANK64=1 (what a text of 2^32 positions or more takes by itself): the saved index is byte for byte the default build's"""
    from moni_align_amd import index_build as ib
    pg = small_case.pg
    a, b = str(tmp_path / "a.mfi"), str(tmp_path / "b.mfi")
    ib.build_from_pangenome(pg, device="cpu").save(a)
    monkeypatch.setenv("MONI_BUILD_RANK64", "1")
    ib.build_from_pangenome(pg, device="cpu").save(b)
    assert open(a, "rb").read() == open(b, "rb").read()
