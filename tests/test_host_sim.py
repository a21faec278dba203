"""The kernels' per-lane code (seed_core.h) replayed on the host over the device index image, against
the oracle: bit-exact pointers, MEMs, halves, occurrence lists and filter counts.  Catches layout and
logic errors without a GPU; the same comparisons run on the real kernels under -m gpu."""
import numpy as np
import pytest

from oracle import orc
from tests.host_sim import sim as hs
from tests.parity import assert_seeds_equal


def ragged(reads_list):
    offs = np.zeros(len(reads_list) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads_list])
    seq = np.concatenate(reads_list) if reads_list else np.zeros(0, np.uint8)
    return seq, offs


@pytest.fixture(scope="module")
def pair(medium_case):
    return orc.OracleIndex(medium_case.path), hs.Sim(medium_case.fi)


def run_both(pair, seq, offs, **kw):
    o, s = pair
    want = o.seed_batch(seq, offs, kw.get("min_len", 25), kw.get("filter_seeds", True), kw.get("n_seeds_thr", 1000))
    got = s.seed_run(seq, offs, **kw)
    return got, want


def test_pointers_and_seeds_150bp(medium_case, pair):
    reads = medium_case.synth.make_reads(medium_case.pg, 300, 150, seed=150)
    seq, offs = ragged(list(reads))
    got, want = run_both(pair, seq, offs)
    o = pair[0]
    rc = medium_case.synth.revcomp(reads)
    for i in range(0, 300, 17):
        assert np.array_equal(got["pointers"][2 * 150 * i:2 * 150 * i + 150], o.ms_query(reads[i].tobytes()))
        assert np.array_equal(got["pointers"][2 * 150 * i + 150:2 * 150 * (i + 1)], o.ms_query(rc[i].tobytes()))
    assert_seeds_equal(got, want)
    assert np.array_equal(got["counters"], want["counters"])
    assert int(got["counters"][0]) == 2 * 150 * 300


def test_ragged_edge_cases(medium_case, pair):
    rng = np.random.default_rng(5)
    base = medium_case.synth.make_reads(medium_case.pg, 40, 250, seed=9, sub_rate=0.03, indel_rate=0.003)
    reads = []
    for i, r in enumerate(base):
        reads.append(r[: int(rng.integers(1, 251))].copy())
    reads.append(np.zeros(0, np.uint8))                       # empty read
    reads.append(np.frombuffer(b"N" * 60, dtype=np.uint8))    # all N (absent from the text)
    reads.append(np.frombuffer(b"acgtacgtacgtacgtacgtacgtacgtacgtacgt", dtype=np.uint8))   # lower case
    x = base[0].copy(); x[40:45] = ord("N"); reads.append(x)
    reads.append(np.frombuffer(bytes(medium_case.fi.text[100:400]), dtype=np.uint8))        # exact 300-mer
    seq, offs = ragged(reads)
    got, want = run_both(pair, seq, offs)
    assert_seeds_equal(got, want)
    assert np.array_equal(got["counters"], want["counters"])


def test_filter_and_small_tmp(medium_case, pair):
    reads = medium_case.synth.make_reads(medium_case.pg, 120, 150, seed=21, sub_rate=0.005)
    seq, offs = ragged(list(reads))
    for thr in (0, 1, 3, 1000):
        got, want = run_both(pair, seq, offs, filter_seeds=True, n_seeds_thr=thr, tmp_cap=2, pool_rows=100000)
        assert_seeds_equal(got, want)
    got, want = run_both(pair, seq, offs, filter_seeds=False, n_seeds_thr=0, tmp_cap=1)
    assert_seeds_equal(got, want)
    got, want = run_both(pair, seq, offs, min_len=12)
    assert_seeds_equal(got, want)


def test_phi_records(medium_case, pair):
    o, s = pair
    rng = np.random.default_rng(1)
    n = medium_case.fi.n
    first = (int(medium_case.fi.ssa[0]) + 1) % n
    last = (int(medium_case.fi.esa[-1]) + 1) % n
    for i in rng.integers(0, n, size=2000):
        i = int(i)
        if i != first:
            assert s.phi(i) == o.phi_lcp(i)
        if i != last:
            assert s.phi(i, True) == o.phi_lcp(i, True)


def long_run_case():
    """A text whose BWT has a run longer than the 12-bit length field, with an LF image that spans thousands of short runs, and a fifth
    letter without a hot slot; reads that cross the junctions.  Returns (flat index, reads)."""
    from moni_align_amd import index_build, synth
    rng = np.random.default_rng(8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    W = acgt[rng.integers(0, 4, size=40)]
    parts = []
    K = 6000
    for k in range(K):
        parts.append(acgt[rng.integers(0, 4, size=12)])          # random left context x_k
        parts.append(np.frombuffer(b"C", dtype=np.uint8))
        parts.append(W)
        parts.append(acgt[rng.integers(0, 4, size=14)])          # distinguishing tail
        if k % 500 == 0:
            parts.append(np.frombuffer(b"NNNN", dtype=np.uint8))  # a fifth letter: N has no hot slot
    seq = np.concatenate(parts)
    pg = synth.Pangenome(seqs=[seq], names=["x"], w=10)
    fi = index_build.build_from_pangenome(pg, device="cpu")
    lens = np.diff(fi.starts.astype(np.int64))
    assert lens.max() >= 4095
    reads = []
    for _ in range(400):                                          # random windows of the text with a few errors
        a = int(rng.integers(0, len(seq) - 150))
        r = seq[a:a + 150].copy()
        for e in rng.integers(0, 150, size=2):
            r[int(e)] = acgt[int(rng.integers(0, 4))]
        reads.append(r)
    return fi, reads


def test_long_run_onto_many_short_runs():
    """A BWT run longer than the 12-bit length field whose LF image spans thousands of short runs: exercises the
    saturated-length check and the galloping search of settle_run, and cold symbols without a hot slot."""
    import tempfile, os
    fi, reads = long_run_case()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "long.mfi")
        fi.save(path)
        o = orc.OracleIndex(path)
        s = hs.Sim(fi)
        sq, offs = ragged(reads)
        got = s.seed_run(sq, offs, min_len=20, n_seeds_thr=100, pool_rows=100000)
        want = o.seed_batch(sq, offs, 20, True, 100)
        assert_seeds_equal(got, want)
        assert np.array_equal(got["counters"], want["counters"])
        for i in range(0, 400, 7):
            assert np.array_equal(got["pointers"][offs[i] * 2:offs[i] * 2 + 150], o.ms_query(reads[i].tobytes()))


def test_no_lcp_form(medium_case):
    """`-n` (seed_finder<slp_t, ms_pointers<>>, seed_finder.hpp:346-370): no sampled LCP - the occurrence walks take Phi / Phi_inv and measure
    the LCP on the text, bounded by the seed's length; the kernels' per-lane code over an image built without slcp against the oracle in that
    mode - and both against the sampled-LCP form, which finds the same seeds on this index"""
    o = orc.OracleIndex(medium_case.path)
    o.set_no_lcp(True)
    s = hs.Sim(medium_case.fi, without_lcp=True)
    reads = list(medium_case.synth.make_reads(medium_case.pg, 400, 150, seed=77, sub_rate=0.02))
    reads.append(np.frombuffer(bytes(medium_case.fi.text[-170:-20]), dtype=np.uint8))          # a read at the very end of the text (the n - curr >= len guard)
    seq, offs = ragged(reads)
    for kw in ({}, {"n_seeds_thr": 2, "pool_rows": 100000}, {"min_len": 12}):
        got, want = run_both((o, s), seq, offs, **kw)
        assert_seeds_equal(got, want)
    o2 = orc.OracleIndex(medium_case.path)
    assert_seeds_equal(s.seed_run(seq, offs), o2.seed_batch(seq, offs, 25, True, 1000))
