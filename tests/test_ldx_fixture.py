"""The reference's own fixture for this path: data/Chr21.10.ldx + .lidx + lifts/*.lft, the files
test/src/lifting_test.cpp:57-141 checks.  Copied as data into tests/golden/ref_data/.  Pins the sdsl serialisation
(sd_vector, int_vector, select_support_mcl), the .ldx layout (liftidx::load) and the lift(pos) semantics:
  * the file is consumed exactly to its last byte,
  * names and sequence starts agree with Chr21.10.lidx (every sequence is followed by w = 10 separators),
  * contig 0 carries a null lift: lift(i) == i                                     (lifting_test.cpp:117-121),
  * a haplotype's lift equals the lift stored in its levioSAM .lft file           (lifting_test.cpp:123-139),
  * zeros(del) = haplotype length and zeros(ins) = reference length, which fixes which vector is which and the
    formula lift_pos(p) = ins.rank0(del.select0(p+1)); the per-file `limits` of the test are exactly zeros(del).
The FASTA-built (null-lift) index used elsewhere in this repo is the contig-0 case."""
import os

import numpy as np

from tests import sdsl_reader as sr

D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_data")
W = 10


def load():
    buf = open(os.path.join(D, "Chr21.10.ldx"), "rb").read()
    return buf, sr.read_ldx_old_layout(buf)


def lidx():
    rows = [l.split() for l in open(os.path.join(D, "Chr21.10.lidx")).read().splitlines() if l.strip()]
    return [r[0] for r in rows], [int(r[1]) for r in rows]


def test_ldx_layout_consumed_exactly_and_matches_lidx():
    buf, d = load()
    names, lens = lidx()
    assert d["consumed"] == len(buf) == 716116
    assert d["names"] == names and len(d["lifts"]) == 9
    onsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    assert np.array_equal(d["starts"].ones, onsets)
    assert d["u"] == int(onsets[-1]) + 1 == d["starts"].size


def test_reference_contig_has_null_lift():
    _, d = load()
    L, second = d["lifts"][0]
    assert second == 0 and L.ins.m == 0 and L.dele.m == 0 and L.snp.m == 0
    names, lens = lidx()
    for p in (0, 1, 12345, lens[0] - W - 1):
        assert second + L.lift_pos(p) == p


def test_haplotype_lift_matches_leviosam_file_and_lengths():
    _, d = load()
    names, lens = lidx()
    lft = open(os.path.join(D, "HG00096_H1_21.lft"), "rb").read()
    c = sr.Cursor(lft)
    assert c.u64() == 1                                   # LiftMap: one contig
    M = sr.Lift(c)
    L, second = d["lifts"][1]
    assert second == 0                                    # lifts onto contig "21", which starts the reference concatenation
    for a, b in ((L.ins, M.ins), (L.dele, M.dele), (L.snp, M.snp)):
        assert np.array_equal(a.ones, b.ones)
    ref_len, hap_len = lens[0] - W, lens[1] - W
    assert M.ins.size - M.ins.m == ref_len                # alignment columns that are not insertions = reference bases
    assert M.dele.size - M.dele.m == hap_len == 46708362  # columns that are not deletions = haplotype bases (= `limits[0]` of the test)
    rng = np.random.default_rng(0)
    ps = np.sort(rng.integers(0, hap_len, size=400))
    lifted = [L.lift_pos(int(p)) for p in ps]
    assert lifted == [M.lift_pos(int(p)) for p in ps]
    assert all(b >= a for a, b in zip(lifted, lifted[1:])) and lifted[0] >= 0 and lifted[-1] < ref_len
    assert L.lift_pos(0) == 0 and L.lift_pos(hap_len - 1) == ref_len - 1
