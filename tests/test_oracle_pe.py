"""The paired-end restatement of the oracle (oracle/align_pe.hpp: the checker of the product's paired path, tests/test_gpu_pe.py;
nothing in the reference tree pins paired-end output).  What can be checked without a reference binary: on simulated FR pairs
with a known insert-size distribution the learnt model is the simulated one, proper pairs carry the SAM invariants (complementary flags,
mirrored TLEN, PNEXT = the mate's POS, RNEXT '='), each mate's alignment is what the single-end path gives for a uniquely placed read,
and the batch order of st_align (learn first, then align the learning batches, then the rest) keeps the records in input order."""
import numpy as np
import pytest

from moni_align_amd import index_build, synth
from oracle import orc


def make_pairs(pg, n, L=100, mean=350.0, sd=30.0, seed=5):
    rng = np.random.default_rng(seed)
    m1, m2, truth = [], [], []
    for i in range(n):
        h = int(rng.integers(0, len(pg.seqs)))
        s = pg.seqs[h]
        ins = int(max(2 * L + 10, rng.normal(mean, sd)))
        p = int(rng.integers(0, len(s) - ins))
        frag = s[p:p + ins].copy()
        k = rng.random(ins) < 0.005
        frag[k] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(k.sum()))]
        a, b = frag[:L].copy(), synth.revcomp(frag[None, ins - L:])[0].copy()
        if rng.random() < 0.5:          # the fragment comes from the other strand: mate 1 is the reverse-strand end
            a, b = b, a
        m1.append(a); m2.append(b); truth.append(ins)
    return m1, m2, truth


@pytest.fixture(scope="module")
def case():
    pg = synth.make_pangenome(80000, 3, site_spacing=800)
    fi = index_build.build_from_pangenome(pg, device="cpu")
    return pg, fi, orc.OracleIndex(fi=fi)


def run(o, m1, m2, b_size=512, slash=True):
    n = len(m1)
    L = len(m1[0])
    offs = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
    nm1 = [("p%d/1" % i if slash else "p%d" % i).encode() for i in range(n)]
    nm2 = [("p%d/2" % i if slash else "q%d" % i).encode() for i in range(n)]
    no1 = np.zeros(n + 1, np.uint64); no1[1:] = np.cumsum([len(x) for x in nm1])
    no2 = np.zeros(n + 1, np.uint64); no2[1:] = np.cumsum([len(x) for x in nm2])
    q = np.full(n * L, ord("I"), np.uint8)
    sam, st = orc.align_pe(o, np.concatenate(m1), offs, np.concatenate(m2), offs, np.frombuffer(b"".join(nm1), np.uint8), no1,
                           np.frombuffer(b"".join(nm2), np.uint8), no2, q, q, b_size=b_size)
    recs = [l.split(b"\t") for l in sam.split(b"\n") if l]
    return recs, st


def test_model_and_sam_invariants(case):
    pg, fi, o = case
    m1, m2, truth = make_pairs(pg, 1500)
    recs, st = run(o, m1, m2)
    assert len(recs) == 2 * 1500
    assert st["ins_complete"] and st["ins_count"] >= 1000
    # `dist` of the model is |pos(mate 2) - (pos(mate 1) + |mate 1|)| in the orientation the chain has (aligner_ksw2.hpp:2175): insert - 2 L
    # when mate 1 is the forward end, the whole insert when it is the reverse end - the reference's measure is not strand-symmetric, so
    # with half of the fragments on either strand the model is the mixture: mean insert - L, variance sd^2 + L^2
    assert abs(st["ins_mean"] - (np.mean(truth) - 100)) < 8 and abs(st["ins_std_dev"] - np.sqrt(30.0 ** 2 + 100.0 ** 2)) < 8
    proper = 0
    for i in range(1500):
        a, b = recs[2 * i], recs[2 * i + 1]
        assert a[0] == b"p%d" % i and b[0] == b"p%d" % i           # /1 and /2 removed (common/sam.hpp:132-141)
        fa, fb = int(a[1]), int(b[1])
        if fa & 2:
            proper += 1
            assert fb & 2 and (fa & 64) and (fb & 128) and not (fa & 4) and not (fb & 4)
            assert bool(fa & 16) == bool(fb & 32) and bool(fa & 32) == bool(fb & 16) and bool(fa & 16) != bool(fb & 16)
            assert a[6] == b"=" and b[6] == b"="
            assert int(a[7]) == int(b[3]) and int(b[7]) == int(a[3])          # PNEXT = the mate's POS
            assert int(a[8]) == -int(b[8]) and int(a[8]) != 0
            assert abs(abs(int(a[8])) - truth[i]) <= 60                        # TLEN ~ the simulated insert (lifted coordinates: indels shift it)
            assert a[2] == b[2]
    assert proper > 1400


def test_mates_agree_with_the_single_end_path(case):
    """a properly paired mate whose single-end alignment is unique (MAPQ 60) gets the same RNAME / POS / CIGAR from both paths"""
    pg, fi, o = case
    m1, m2, _ = make_pairs(pg, 300, seed=9)
    recs, _ = run(o, m1, m2)
    L = 100
    offs = np.arange(0, 301 * L, L, dtype=np.uint64)
    names, noff = orc.make_names(300)
    se1, _ = orc.align_batch(o, np.concatenate(m1), offs, names, noff, np.full(300 * L, ord("I"), np.uint8))
    se = [l.split(b"\t") for l in se1.split(b"\n") if l]
    same = checked = 0
    for i in range(300):
        a = recs[2 * i]
        if int(a[1]) & 2 and int(se[i][4]) == 60:
            checked += 1
            same += (a[2], a[3], a[5]) == (se[i][2], se[i][3], se[i][5])
    assert checked > 200 and same == checked


def test_batch_order_and_small_inputs(case):
    pg, fi, o = case
    m1, m2, _ = make_pairs(pg, 130, seed=3)
    recs, st = run(o, m1, m2, b_size=50, slash=False)          # never reaches 1000 confident pairs: the model of the batches seen is used
    assert not st["ins_complete"] and 0 < st["ins_count"] <= 130
    assert [r[0] for r in recs[0::2]] == [b"p%d" % i for i in range(130)]
    assert [r[0] for r in recs[1::2]] == [b"q%d" % i for i in range(130)]
    # different names: RNEXT is the mate's name (aligner_ksw2.hpp:733-742)
    assert all(r[6] in (b"q%d" % i, b"=") for i, r in enumerate(recs[0::2]))
