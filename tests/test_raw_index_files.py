"""Reader of the reference builder's raw per-run files (§8(f) item 1): 5-byte little-endian arrays,
`SA ? SA - 1 : n - 1` sample rule (moni.hpp:148-184), F as build_F_ (moni.hpp:253-282), `.lidx` as the reference's own
fixture spells it.  Pinned by the loading code cited in index_build.py and a write/read round trip of a built index; no
upstream-built raw files exist here."""
import os

import numpy as np
import pytest

from moni_align_amd import index_build


def test_five_byte_little_endian(tmp_path):
    p = str(tmp_path / "x")
    vals = np.array([0, 1, 255, 256, (1 << 40) - 1, 0x0102030405], dtype=np.uint64)
    index_build._write5(p, vals)
    raw = open(p, "rb").read()
    assert len(raw) == 5 * len(vals) and raw[25:30] == bytes([5, 4, 3, 2, 1])
    assert np.array_equal(index_build._read5(p), vals)
    with pytest.raises(ValueError):
        index_build._write5(p, np.array([1 << 40], dtype=np.uint64))
    open(p, "wb").write(b"1234")
    with pytest.raises(ValueError):
        index_build._read5(p)


def test_lidx_of_the_reference_fixture():
    d = os.path.join(os.path.dirname(__file__), "golden", "ref_data")
    names, on = index_build.read_lidx(os.path.join(d, "Chr21.10.lidx"), 10)
    assert names[0] == "21" and names[1] == "HG00096_H1_21" and len(names) == 9
    assert int(on[1]) == 46709993 and int(on[2]) == 46709993 + 46708372


def test_round_trip_and_sample_rule(small_case, tmp_path):
    fi = small_case.fi
    pre = str(tmp_path / "idx")
    index_build.to_raw_files(fi, pre, pre + ".txt", pre + ".lidx")
    # the run whose SA sample is 0 is stored as 0 and read back as n - 1
    esa_pairs = index_build._read5(pre + ".esa")
    k = int(np.nonzero(fi.esa == fi.n - 1)[0][0]) if (fi.esa == fi.n - 1).any() else None
    if k is not None:
        assert esa_pairs[2 * k + 1] == 0
    assert os.path.getsize(pre + ".bwt.len") == 5 * fi.r and os.path.getsize(pre + ".ssa") == 10 * fi.r
    got = index_build.from_raw_files(pre, pre + ".txt", pre + ".lidx", fi.w)
    assert got.n == fi.n and got.r == fi.r and got.names == fi.names
    for f in ("F", "heads", "starts", "ssa", "esa", "thr", "slcp", "text", "seq_starts"):
        assert np.array_equal(getattr(got, f), getattr(fi, f)), f
    # the CLI form
    index_build._main([pre, "--text", pre + ".txt", "--lidx", pre + ".lidx", "-w", str(fi.w), "-o", pre + ".mfi"])
    again = index_build.FlatIndex.load(pre + ".mfi")
    assert np.array_equal(again.ssa, fi.ssa) and again.names == fi.names


def test_rejects_inconsistent_files(small_case, tmp_path):
    fi = small_case.fi
    pre = str(tmp_path / "idx")
    index_build.to_raw_files(fi, pre, pre + ".txt", pre + ".lidx")
    open(pre + ".thr_pos", "ab").write(b"\0" * 5)
    with pytest.raises(ValueError):
        index_build.from_raw_files(pre, pre + ".txt", pre + ".lidx", fi.w)
