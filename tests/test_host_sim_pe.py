"""Paired-end, CPU side: the per-pair logic the pe_align_kernel lanes run (pe_core.h) replayed on the host with the oracle's ksw2 standing
in for the DP kernels, plus the host finishing (pe_host.hpp), against the oracle's restatement (oracle/align_pe.hpp): SAM text must be
byte-identical, and the learn pass must reproduce the oracle's fragment model bit for bit.  The same comparison runs with the real
kernels under -m gpu (tests/test_gpu_pe.py)."""
import math

import numpy as np
import pytest

from moni_align_amd import index_build, synth
from oracle import orc
from tests.host_sim import sim as hs
from tests.test_oracle_pe import make_pairs


def interleave(m1, m2, slash=True, names=None):
    n = len(m1)
    reads = [x for p in zip(m1, m2) for x in p]
    offs = np.zeros(2 * n + 1, np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    nm = names or [(("p%d/%d" % (i, k + 1)) if slash else ("%s%d" % ("pq"[k], i))).encode() for i in range(n) for k in range(2)]
    noff = np.zeros(2 * n + 1, np.uint64)
    noff[1:] = np.cumsum([len(x) for x in nm])
    seq = np.concatenate(reads)
    return seq, offs, np.frombuffer(b"".join(nm), np.uint8), noff, np.full(len(seq), ord("I"), np.uint8)


def oracle_pe(o, m1, m2, slash=True, b_size=512, find_orphan=False, report_mems=False, secondary_chains=False, csv=False, filter_dir=True):
    n = len(m1)
    o1 = np.zeros(n + 1, np.uint64); o1[1:] = np.cumsum([len(x) for x in m1])
    o2 = np.zeros(n + 1, np.uint64); o2[1:] = np.cumsum([len(x) for x in m2])
    nm1 = [("p%d/1" % i if slash else "p%d" % i).encode() for i in range(n)]
    nm2 = [("p%d/2" % i if slash else "q%d" % i).encode() for i in range(n)]
    no1 = np.zeros(n + 1, np.uint64); no1[1:] = np.cumsum([len(x) for x in nm1])
    no2 = np.zeros(n + 1, np.uint64); no2[1:] = np.cumsum([len(x) for x in nm2])
    q1 = np.full(int(o1[-1]), ord("I"), np.uint8); q2 = np.full(int(o2[-1]), ord("I"), np.uint8)
    return orc.align_pe(o, np.concatenate(m1), o1, np.concatenate(m2), o2, np.frombuffer(b"".join(nm1), np.uint8), no1,
                        np.frombuffer(b"".join(nm2), np.uint8), no2, q1, q2, b_size=b_size, find_orphan=find_orphan, report_mems=report_mems, secondary_chains=secondary_chains, csv=csv, filter_dir=filter_dir)


def first_diff(a: bytes, b: bytes):
    la, lb = a.split(b"\n"), b.split(b"\n")
    for k, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return k, x.decode()[:500], y.decode()[:500]
    return min(len(la), len(lb)), "<len %d>" % len(la), "<len %d>" % len(lb)


def welford(learn, min_score):
    """learn_fragment_model over one batch (aligner_ksw2.hpp:816-885), gap threshold 0."""
    count, mean, m2 = 0, 0.0, 0.0
    for (al, tot, s2, dist), ms in zip(learn, min_score):
        if not al:
            continue
        if s2 >= ms and not (int(tot) - int(s2) > 0):
            continue
        value = float(dist)
        delta = value - mean
        count += 1
        mean += delta / count
        m2 += delta * (value - mean)
    return count, mean, m2


@pytest.fixture(scope="module")
def case():
    pg = synth.make_pangenome(80000, 3, site_spacing=800)
    fi = index_build.build_from_pangenome(pg, device="cpu")
    return pg, fi, orc.OracleIndex(fi=fi)


def hard_pairs(pg, n=600, seed=23):
    """pairs with noise, one mate unalignable, both unalignable, mates on different haplotypes / far apart, ragged lengths"""
    rng = np.random.default_rng(seed)
    m1, m2, _ = make_pairs(pg, n, L=100, seed=seed)
    for i in range(0, n, 7):            # mate 2 is noise
        m2[i] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=100)]
    for i in range(3, n, 11):           # mate 1 is noise
        m1[i] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=100)]
    for i in range(5, n, 13):           # both
        m1[i] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=100)]
        m2[i] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=100)]
    for i in range(2, n, 17):           # improper: mate 2 from somewhere else
        j = (i * 31 + 7) % n
        m2[i] = m2[j].copy()
    for i in range(4, n, 9):            # ragged
        m1[i] = m1[i][: int(rng.integers(40, 100))].copy()
        m2[i] = m2[i][: int(rng.integers(40, 100))].copy()
    for i in range(1, n, 23):           # one mate keeps a 27-base seed and is noise otherwise: the chain pairs, the mate's score stays under its minimum
        keep = m2[i][36:63].copy()
        m2[i] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=len(m2[i]))].copy()
        m2[i][36:63] = keep
    for i in range(8, n, 29):
        keep = m1[i][30:57].copy()
        m1[i] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=len(m1[i]))].copy()
        m1[i][30:57] = keep
    for i in range(6, n, 19):           # heavy substitutions
        k = rng.random(len(m1[i])) < 0.06
        m1[i] = m1[i].copy(); m1[i][k] = ord("A")
    return m1, m2


def test_pe_replay_matches_oracle(case):
    pg, fi, o = case
    m1, m2, _ = make_pairs(pg, 1300)
    want, st = oracle_pe(o, m1, m2, b_size=512)
    assert st["ins_complete"]
    seq, offs, names, noff, q = interleave(m1, m2)
    S = hs.Sim(fi)
    # the learn pass over the batches the oracle learnt on (512 + 512 pairs reach the 1000 the model wants)
    tot_cnt, mean, m2acc = 0, 0.0, 0.0
    for b in range(2):
        lo, hi = 512 * b, min(512 * (b + 1), len(m1))
        sl = slice(2 * lo, 2 * hi + 1)
        learn, _ = S.align_pe_batch(seq[int(offs[2 * lo]):int(offs[2 * hi])], offs[sl] - offs[2 * lo], names[int(noff[2 * lo]):int(noff[2 * hi])],
                                    noff[sl] - noff[2 * lo], q[int(offs[2 * lo]):int(offs[2 * hi])], finalize=False)
        msc = [int(20 + 8 * math.log(len(m1[i]))) + int(20 + 8 * math.log(len(m2[i]))) for i in range(lo, hi)]
        c, mu, mm = welford(learn, msc)
        if tot_cnt:
            t = tot_cnt + c
            d = mean - mu
            m2acc += mm + (d * d * tot_cnt * c) / t
            mean = (tot_cnt * mean + c * mu) / t
            tot_cnt = t
        else:
            tot_cnt, mean, m2acc = c, mu, mm
    assert tot_cnt == st["ins_count"] and mean == st["ins_mean"] and math.sqrt(m2acc / tot_cnt) == st["ins_std_dev"]
    got, stats = S.align_pe_batch(seq, offs, names, noff, q, finalize=True, mean=st["ins_mean"], std_dev=st["ins_std_dev"])
    assert int(stats[3]) == 0, "pairs overflowed the device capacities"
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))
    assert int(stats[1]) == st["aligned"] and st["aligned"] > 1200


def test_pe_replay_hard_cases(case):
    pg, fi, o = case
    m1, m2 = hard_pairs(pg)
    want, st = oracle_pe(o, m1, m2, slash=False, b_size=4096)
    seq, offs, names, noff, q = interleave(m1, m2, slash=False)
    got, stats = hs.Sim(fi).align_pe_batch(seq, offs, names, noff, q, finalize=True, mean=st["ins_mean"], std_dev=st["ins_std_dev"])
    assert int(stats[3]) == 0
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))
    flags = [int(l.split(b"\t")[1]) for l in want.split(b"\n") if l]
    assert any(f & 8 and not f & 4 for f in flags) and any(f & 4 and f & 1 for f in flags) and any(f == 4 for f in flags) and any(f & 2 for f in flags)


def seedless_pairs(pg, n=600, seed=5, periods=(19, 4, 5, 19, 6, 3)):
    """every 6th pair: mate 2 (or mate 1) gets a substitution every 19 bases - no 25-base MEM, the pair fails jointly and is a case for orphan
    recovery; a few pairs have one mate of noise (orphan search finds nothing good enough)"""
    rng = np.random.default_rng(seed)
    m1, m2, _ = make_pairs(pg, n, seed=seed)
    for i in range(5, n, 6):
        tgt = m2 if (i // 6) % 2 == 0 else m1
        x = tgt[i].copy()
        # the period sets the recovered mate's global score against its minimum 20 + 8 ln(len): 19 far above, 6 and 5 above, 4 and 3 below -
        # recovery then finds the place but the mate is reported unmapped beside its mapped partner (aligner_ksw2.hpp:2471,2519)
        step = periods[(i // 6) % len(periods)]
        for p in range(9 if step == 19 else 2, len(x), step):
            x[p] = ord("A") if x[p] != ord("A") else ord("C")
        tgt[i] = x
    for i in range(2, n, 37):
        m2[i] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=len(m2[i]))].copy()
    return m1, m2


def oracle_pe_orphan(o, m1, m2, b_size=512):
    n = len(m1)
    o1 = np.zeros(n + 1, np.uint64); o1[1:] = np.cumsum([len(x) for x in m1])
    o2 = np.zeros(n + 1, np.uint64); o2[1:] = np.cumsum([len(x) for x in m2])
    nm1 = [b"p%d/1" % i for i in range(n)]; nm2 = [b"p%d/2" % i for i in range(n)]
    no1 = np.zeros(n + 1, np.uint64); no1[1:] = np.cumsum([len(x) for x in nm1])
    no2 = np.zeros(n + 1, np.uint64); no2[1:] = np.cumsum([len(x) for x in nm2])
    q1 = np.full(int(o1[-1]), ord("I"), np.uint8); q2 = np.full(int(o2[-1]), ord("I"), np.uint8)
    return orc.align_pe(o, np.concatenate(m1), o1, np.concatenate(m2), o2, np.frombuffer(b"".join(nm1), np.uint8), no1,
                        np.frombuffer(b"".join(nm2), np.uint8), no2, q1, q2, b_size=b_size, find_orphan=True)


def test_pe_replay_orphan_recovery(case):
    pg, fi, o = case
    m1, m2 = seedless_pairs(pg)
    want, st = oracle_pe_orphan(o, m1, m2, b_size=4096)
    assert st["orphan_recovered"] > 20 and st["orphan_pairs"] > st["orphan_recovered"]
    seq, offs, names, noff, q = interleave(m1, m2)
    got, stats = hs.Sim(fi).align_pe_batch(seq, offs, names, noff, q, finalize=True, mean=st["ins_mean"], std_dev=st["ins_std_dev"], find_orphan=True)
    assert int(stats[3]) == 0
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))
    assert int(stats[1]) == st["aligned"]
    assert_orphan_flags(want)


def assert_orphan_flags(sam):
    """both outcomes of a recovery are present: the recovered mate above its minimum (a proper pair whose mate carries no ZS of its own is not
    distinguishable by flag, so count pairs) and below it (flags 73 / 133 or 137 / 69: mapped mate + unmapped recovered mate)"""
    flags = [int(l.split(b"\t")[1]) for l in sam.split(b"\n") if l]
    pairs = set(zip(flags[0::2], flags[1::2]))
    assert {(73, 133), (89, 133)} & pairs and {(69, 137), (69, 153)} & pairs, sorted(pairs)


def test_pe_host_pipeline_replay(case):
    """pe_big.cpp (the host pipeline for pairs: pe_core.h compiled with large capacities in its own namespace) on the CPU, every pair through
    it: hard cases, and orphan recovery - its records must give the oracle's text too"""
    pg, fi, o = case
    S = hs.Sim(fi)
    m1, m2 = hard_pairs(pg, n=300, seed=29)
    want, st = oracle_pe(o, m1, m2, slash=False, b_size=4096)
    seq, offs, names, noff, q = interleave(m1, m2, slash=False)
    got, stats = S.align_pe_big_batch(seq, offs, names, noff, q, mean=st["ins_mean"], std_dev=st["ins_std_dev"])
    assert int(stats[3]) == 0
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))
    m1, m2 = seedless_pairs(pg, n=300, seed=8)
    want, st = oracle_pe_orphan(o, m1, m2, b_size=4096)
    assert st["orphan_recovered"] > 10
    seq, offs, names, noff, q = interleave(m1, m2)
    got, stats = S.align_pe_big_batch(seq, offs, names, noff, q, mean=st["ins_mean"], std_dev=st["ins_std_dev"], find_orphan=True)
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))
    assert int(stats[1]) == st["aligned"]


def test_pe_replay_secondary_chains(case):
    """-Z (find_chains_secondary, chain.hpp:442-727): the second track of the chaining in pe_core.h's lane-serial form against the oracle's restatement,
    with orphan recovery on; the option changes most records (more chains: sub_n, MAPQ, alternatives)"""
    pg, fi, o = case
    m1, m2, _ = make_pairs(pg, 500)
    h1, h2 = hard_pairs(pg)
    m1, m2 = list(m1) + list(h1), list(m2) + list(h2)
    want, st = oracle_pe(o, m1, m2, b_size=4096, find_orphan=True, secondary_chains=True)
    plain, _ = oracle_pe(o, m1, m2, b_size=4096, find_orphan=True)
    assert sum(x != y for x, y in zip(want.split(b"\n"), plain.split(b"\n"))) > 100
    seq, offs, names, noff, q = interleave(m1, m2)
    got, stats = hs.Sim(fi).align_pe_batch(seq, offs, names, noff, q, finalize=True, mean=st["ins_mean"], std_dev=st["ins_std_dev"], find_orphan=True, secondary_chains=True)
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))
    assert int(stats[1]) == st["aligned"]
