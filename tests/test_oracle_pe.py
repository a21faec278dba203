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


def run(o, m1, m2, b_size=512, slash=True, **kw):
    n = len(m1)
    L = len(m1[0])
    offs = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
    nm1 = [("p%d/1" % i if slash else "p%d" % i).encode() for i in range(n)]
    nm2 = [("p%d/2" % i if slash else "q%d" % i).encode() for i in range(n)]
    no1 = np.zeros(n + 1, np.uint64); no1[1:] = np.cumsum([len(x) for x in nm1])
    no2 = np.zeros(n + 1, np.uint64); no2[1:] = np.cumsum([len(x) for x in nm2])
    q = np.full(n * L, ord("I"), np.uint8)
    sam, st = orc.align_pe(o, np.concatenate(m1), offs, np.concatenate(m2), offs, np.frombuffer(b"".join(nm1), np.uint8), no1,
                           np.frombuffer(b"".join(nm2), np.uint8), no2, q, q, b_size=b_size, **kw)
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


def _naive_local(q, t, mat, gapo, gape):
    """Gotoh local alignment with full matrices: best score and, for every cell, H (independent of the rolling-row restatement)"""
    n, m = len(t), len(q)
    H = np.zeros((n + 1, m + 1), np.int64); E = np.zeros((n + 1, m + 1), np.int64); F = np.zeros((n + 1, m + 1), np.int64)
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            E[i, j] = max(E[i - 1, j] - gape, H[i - 1, j] - gapo - gape, 0)
            F[i, j] = max(F[i, j - 1] - gape, H[i, j - 1] - gapo - gape, 0)
            H[i, j] = max(0, H[i - 1, j - 1] + int(mat[int(t[i - 1]) * 5 + int(q[j - 1])]), E[i, j], F[i, j])
    return H


def test_ksw_align_restatement_against_full_matrices():
    """klib's ksw_align as restated for fill_orphan: the score is the full-matrix optimum, (te, qe) is the first row reaching it and the
    leftmost query position in that row, and the reported start (tb, qb) bounds a segment whose best local score is the same"""
    rng = np.random.default_rng(3)
    mat = orc.DEFAULT_MAT
    for trial in range(40):
        m, n = int(rng.integers(20, 60)), int(rng.integers(40, 160))
        t = rng.integers(0, 4, size=n).astype(np.uint8)
        at = int(rng.integers(0, n - 15))
        q = t[at:at + m].copy()[:m]
        k = rng.random(len(q)) < 0.12
        q[k] = (q[k] + 1) & 3
        if trial % 5 == 0 and len(q) > 12:
            q = np.concatenate([q[:6], q[9:]])          # a deletion in the query
        if trial % 7 == 0:
            t[int(rng.integers(0, n))] = 4              # an N in the target scores 0
        r = orc.ksw_align(q, t)
        H = _naive_local(q, t, mat, 4, 2)
        assert r["score"] == int(H.max())
        rows = np.where(H.max(axis=1) == H.max())[0]
        assert r["te"] == int(rows[0]) - 1
        assert r["qe"] == int(np.where(H[rows[0]] == H.max())[0][0]) - 1
        assert 0 <= r["tb"] <= r["te"] and 0 <= r["qb"] <= r["qe"]
        sub = _naive_local(q[r["qb"]:r["qe"] + 1], t[r["tb"]:r["te"] + 1], mat, 4, 2)
        assert int(sub.max()) == r["score"] and int(sub[-1, -1]) == r["score"]      # the segment ends at its last cell with the whole score


def test_orphan_recovery_places_the_seedless_mate(case):
    """a mate with a substitution every 19 bases has no 25-base MEM: the pair fails jointly; with find_orphan the mate is found by local
    alignment inside the window the fragment model predicts, and the pair comes out proper at the simulated position"""
    pg, fi, o = case
    m1, m2, truth = make_pairs(pg, 1200)
    broken = list(range(7, 1200, 10))
    for i in broken:
        x = m2[i].copy()
        for p in range(9, len(x), 19):
            x[p] = ord("A") if x[p] != ord("A") else ord("C")
        m2[i] = x
    n, L = len(m1), 100
    offs = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
    nm1 = [b"p%d/1" % i for i in range(n)]; nm2 = [b"p%d/2" % i for i in range(n)]
    no1 = np.zeros(n + 1, np.uint64); no1[1:] = np.cumsum([len(x) for x in nm1])
    no2 = np.zeros(n + 1, np.uint64); no2[1:] = np.cumsum([len(x) for x in nm2])
    q = np.full(n * L, ord("I"), np.uint8)
    args = (o, np.concatenate(m1), offs, np.concatenate(m2), offs, np.frombuffer(b"".join(nm1), np.uint8), no1, np.frombuffer(b"".join(nm2), np.uint8), no2, q, q)
    off_sam, off_st = orc.align_pe(*args, b_size=512, find_orphan=False)
    on_sam, on_st = orc.align_pe(*args, b_size=512, find_orphan=True)
    ro = [l.split(b"\t") for l in off_sam.split(b"\n") if l]
    rn = [l.split(b"\t") for l in on_sam.split(b"\n") if l]
    assert on_st["ins_mean"] == off_st["ins_mean"] and on_st["orphan_pairs"] == off_st["orphan_pairs"] >= len(broken) - 5
    # the search window lies to the right of the anchored mate when that is mate 1 and to the left when it is mate 2 (aligner_ksw2.hpp:2398-2420),
    # whatever the strand: a fragment whose mate 1 is the reverse-strand end has its mate 2 on the other side and is not recovered - half of these
    assert off_st["orphan_recovered"] == 0 and on_st["orphan_recovered"] >= len(broken) // 3
    rec = 0
    for i in range(n):
        if i not in set(broken):
            if int(ro[2 * i][1]) != 4:                                               # pairs that align jointly are untouched
                assert ro[2 * i] == rn[2 * i] and ro[2 * i + 1] == rn[2 * i + 1]
            continue
        a, b = rn[2 * i], rn[2 * i + 1]
        if int(a[1]) & 2:
            rec += 1
            assert int(b[1]) & 2 and int(a[7]) == int(b[3]) and int(b[7]) == int(a[3]) and int(a[8]) == -int(b[8])
            assert abs(abs(int(a[8])) - truth[i]) <= 60
            assert b"M" in b[5] and int(dict(x.split(b":", 2)[::2] for x in b[11:] if x.startswith(b"NM"))[b"NM"]) >= 4
    assert rec >= len(broken) // 3


def test_secondary_chains_keep_the_pair_invariants_and_the_placements(case):
    """-Z (find_chains_secondary, chain.hpp:442-727): more chains reach the selection loop, so second-best scores, sub_n, MAPQ and the alternative hits
    change - but a pair's placement does not get worse: the same pairs are proper, at the same positions, with the SAM pair invariants, and no MAPQ rises"""
    pg, fi, o = case
    m1, m2, truth = make_pairs(pg, 800, seed=4)
    plain, st0 = run(o, m1, m2)
    sec, st1 = run(o, m1, m2, secondary_chains=True)
    assert len(sec) == len(plain) == 1600
    changed = sum(1 for x, y in zip(plain, sec) if x != y)
    assert changed > 50
    proper = 0
    for i in range(800):
        a, b = sec[2 * i], sec[2 * i + 1]
        pa, pb = plain[2 * i], plain[2 * i + 1]
        fa, fb = int(a[1]), int(b[1])
        if int(pa[1]) & 2:
            assert fa & 2 and fb & 2
            assert (a[2], a[3], a[5]) == (pa[2], pa[3], pa[5]) and (b[2], b[3], b[5]) == (pb[2], pb[3], pb[5])          # RNAME, POS, CIGAR
            assert int(a[4]) <= int(pa[4]) and int(b[4]) <= int(pb[4])                                                    # MAPQ: more competitors, never fewer
        if fa & 2:
            proper += 1
            assert (fa & 64) and (fb & 128) and bool(fa & 16) != bool(fb & 16) and a[6] == b"=" and b[6] == b"="
            assert int(a[7]) == int(b[3]) and int(b[7]) == int(a[3]) and int(a[8]) == -int(b[8])
    assert proper > 740 and st1["aligned"] >= st0["aligned"] - 2
