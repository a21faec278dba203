"""Lift-over (liftidx::lift / lift_cigar over levioSAM's lift::Lift; aligner_ksw2.hpp:565-576, 444, 3133-3175).

Three implementations are compared here: the oracle's naive rank/select restatement (oracle/flat_index.hpp), the pure-Python
sd_vector reader of tests/sdsl_reader.py, and the product's column-run tables (lift_build.hpp + lift_core.h, through the
host_sim harness; the same header is compiled into the kernels).  Pins:
  * lift(pos) of all 8 haplotypes of the reference's fixture data/Chr21.10.ldx against the levioSAM maps data/lifts/*.lft
    (test/src/lifting_test.cpp:57-141), null lift on contig 0, the linear continuation past a haplotype's end that test expects;
  * lift_cigar: product == oracle on random CIGARs (real fixture lifts and synthetic ones), and a check that shares no code
    with either: a haplotype substring lifted to the reference differs from it exactly at the haplotype's SNPs.
lift_cigar's semantics themselves are [UPSTREAM-RECALL] (levioSAM is an absent submodule): parity with upstream unpinned."""
import os

import numpy as np
import pytest

from moni_align_amd import index_build, synth
from oracle import orc
from tests import sdsl_reader as sr
from tests.host_sim import sim as hs

D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_data")
W = 10
LFT = ["HG00096_H1_21", "HG00096_H2_21", "HG00097_H1_21", "HG00097_H2_21", "HG00099_H1_21", "HG00099_H2_21", "HG00100_H1_21", "HG00100_H2_21"]


@pytest.fixture(scope="module")
def fixture_lifts():
    buf = open(os.path.join(D, "Chr21.10.ldx"), "rb").read()
    d = sr.read_ldx_old_layout(buf)
    lf = index_build.Lifts.from_lists([s for _, s in d["lifts"]], [L.ins.size for L, _ in d["lifts"]],
                                      [L.ins.ones for L, _ in d["lifts"]], [L.dele.ones for L, _ in d["lifts"]])
    starts = d["starts"].ones.astype(np.uint64)
    return d, lf, starts, hs.LiftSim(starts, W, lf)


def test_fixture_all_eight_lifts_match_leviosam_maps(fixture_lifts):
    d, lf, starts, prod = fixture_lifts
    assert d["names"][1:] == LFT
    rng = np.random.default_rng(3)
    # contig 0: identity (lifting_test.cpp:117-121)
    p0 = rng.integers(0, int(starts[1]) - W, size=2000).astype(np.uint64)
    assert np.array_equal(prod.lift(p0), p0)
    for j, name in enumerate(LFT, start=1):
        c = sr.Cursor(open(os.path.join(D, name + ".lft"), "rb").read())
        assert c.u64() == 1
        M = sr.Lift(c)
        L, second = d["lifts"][j]
        assert second == 0
        assert np.array_equal(L.ins.ones, M.ins.ones) and np.array_equal(L.dele.ones, M.dele.ones)
        hap_len = int(starts[j + 1] - starts[j]) - W
        zeros = M.dele.size - M.dele.m         # == hap_len except where levioSAM's construction is off by a few ("buggy", lifting_test.cpp:96)
        assert abs(zeros - hap_len) < 64
        # sampled positions + every position around the first indel events
        ev = np.unique(np.concatenate([M.ins.ones[:40], M.dele.ones[:40]]).astype(np.int64))
        lim = min(zeros, hap_len)
        near = np.unique(np.clip(np.array([M.dele.rank0(int(e)) + k for e in ev for k in range(-3, 4)]), 0, lim - 1))
        ps = np.unique(np.concatenate([rng.integers(0, lim, size=1500), near, [0, lim - 1]])).astype(np.uint64)
        want = np.array([M.lift_pos(int(p)) for p in ps], dtype=np.uint64)
        assert np.array_equal(prod.lift(ps + starts[j]), want), name
        # past the last alignment column (the w separator bytes): sd_vector's select_0 / rank_1 continue linearly there, which is
        # what lifting_test.cpp:133-134 expects of the two haplotypes whose `limits` entry is their own length
        tail = np.arange(zeros, hap_len + W, dtype=np.uint64)
        assert np.array_equal(prod.lift(tail + starts[j]), tail + np.uint64(M.dele.m) - np.uint64(M.ins.m)), name
        if name in ("HG00097_H1_21", "HG00097_H2_21"):
            assert zeros == hap_len
            last = M.lift_pos(hap_len - 1)
            t2 = np.arange(hap_len, hap_len + W, dtype=np.uint64)
            assert np.array_equal(prod.lift(t2 + starts[j]), (t2 - np.uint64(hap_len) + np.uint64(1) + np.uint64(last))), name
        assert prod.n_runs(j) >= 3


def test_oracle_lift_matches_fixture_maps(fixture_lifts, tmp_path):
    d, lf, starts, prod = fixture_lifts
    # an oracle index needs an r-index; give it a tiny one and only exercise its lifts through a direct handle
    pg = synth.make_pangenome(300, 0)
    fi = index_build.build_from_pangenome(pg, device="cpu", lifted=False)
    fi.seq_starts = starts.copy()
    fi.names = list(d["names"])
    fi.lifts = lf
    o = orc.OracleIndex(fi=fi)
    rng = np.random.default_rng(4)
    for j in (1, 4, 8):
        hap_len = int(starts[j + 1] - starts[j]) - W
        ps = rng.integers(0, hap_len, size=400).astype(np.uint64) + starts[j]
        got = np.array([o.lift(int(p)) for p in ps], dtype=np.uint64)
        assert np.array_equal(got, prod.lift(ps))


def random_cigar(rng, n_ops):
    ops = []
    prev = -1
    for _ in range(n_ops):
        op = int(rng.choice([0, 0, 0, 1, 2]))
        if op == prev:
            op = 0 if op else 1
        ln = int(rng.integers(0, 4)) if rng.random() < 0.1 else int(rng.integers(1, 60))      # zero-length operations occur (aligner_ksw2.hpp:2939)
        ops.append((ln << 4) | op)
        prev = op
    return np.array(ops, dtype=np.uint32)


def test_lift_cigar_product_equals_oracle_on_fixture_and_synthetic(fixture_lifts):
    d, lf, starts, prod = fixture_lifts
    pg0 = synth.make_pangenome(300, 0)
    fi = index_build.build_from_pangenome(pg0, device="cpu", lifted=False)
    fi.seq_starts = starts.copy(); fi.names = list(d["names"]); fi.lifts = lf
    o = orc.OracleIndex(fi=fi)
    rng = np.random.default_rng(5)
    for j in range(0, 9):
        L, _ = d["lifts"][j]
        hap_len = int(starts[j + 1] - starts[j]) - W
        ev = np.concatenate([L.ins.ones[:200], L.dele.ones[:200]]).astype(np.int64)
        for t in range(60):
            if len(ev) and t % 2 == 0:       # start close to an indel event so that the walk crosses it
                col = int(rng.choice(ev))
                p = max(0, min(hap_len - 400, L.dele.rank0(col) - int(rng.integers(0, 120))))
            else:
                p = int(rng.integers(0, hap_len - 400))
            cg = random_cigar(rng, int(rng.integers(1, 9)))
            a = prod.lift_cigar(int(starts[j]) + p, cg)
            b = o.lift_cigar(cg, int(starts[j]) + p)
            assert np.array_equal(a, b), (j, p, cg, a, b)
            # read bases are preserved
            q = lambda c: int(sum(x >> 4 for x in c if (x & 15) in (0, 1)))
            assert q(a) == q(cg)
    # dense synthetic lifts: every read-length window crosses events
    pg = synth.make_pangenome(20000, 4, site_spacing=90)
    fl = index_build.build_from_pangenome(pg, device="cpu")
    o2 = orc.OracleIndex(fi=fl)
    p2 = hs.LiftSim(fl.seq_starts, fl.w, fl.lifts)
    for j in range(1, 5):
        for _ in range(200):
            p = int(rng.integers(0, len(pg.seqs[j]) - 500))
            cg = random_cigar(rng, int(rng.integers(1, 7)))
            assert np.array_equal(p2.lift_cigar(int(fl.seq_starts[j]) + p, cg), o2.lift_cigar(cg, int(fl.seq_starts[j]) + p))


def test_lifted_haplotype_substring_differs_from_reference_only_at_snps():
    """No code shared with the restatements: a haplotype substring (CIGAR = its length in M) lifted to the reference must
    spell an alignment whose M columns mismatch exactly at the haplotype's SNP sites, whose I / D columns are the
    haplotype's inserted / deleted bases, and which starts at lift(pos)."""
    pg = synth.make_pangenome(30000, 5, site_spacing=120)
    fl = index_build.build_from_pangenome(pg, device="cpu")
    prod = hs.LiftSim(fl.seq_starts, fl.w, fl.lifts)
    ref = pg.seqs[0]
    rng = np.random.default_rng(6)
    for j in range(1, 6):
        hap = pg.seqs[j]
        vp, vk, vl = pg.variants[j - 1]
        snp_ref = set(int(p) for p, k in zip(vp, vk) if k == 0)
        for _ in range(300):
            ln = int(rng.integers(20, 400))
            p = int(rng.integers(0, len(hap) - ln))
            cg = prod.lift_cigar(int(fl.seq_starts[j]) + p, np.array([ln << 4], dtype=np.uint32))
            t = int(prod.lift(np.array([int(fl.seq_starts[j]) + p], dtype=np.uint64))[0])
            q = 0
            for c in cg:
                op, l = int(c) & 15, int(c) >> 4
                if op == 0:
                    mism = np.nonzero(hap[p + q:p + q + l] != ref[t:t + l])[0]
                    assert set(int(t + x) for x in mism) <= snp_ref
                    assert all((int(t + x) in snp_ref) == (hap[p + q + x] != ref[t + x]) for x in range(l) if int(t + x) in snp_ref)
                    q += l; t += l
                elif op == 1:
                    q += l
                else:
                    t += l
            assert q == ln
