"""The reference's on-disk index pieces through the product's C++ readers / writers (sdsl_io.hpp, ref_index_io.hpp) and the C ABI.

  * data/Chr21.10.ldx (the reference's own fixture, older layout without `w`) loads, and WRITING IT AGAIN GIVES THE SAME BYTES:
    int_vector, bit_vector, sd_vector and both select_support_mcl of all 28 sd_vectors are exactly what sdsl serialised, so an .ldx
    written here loads in the reference.  The current layout (with `w`) is 8 bytes longer and loads back.
  * -m gpu: liftidx::lift of positions of all 8 haplotypes on the GPU (moni_ldx_lift_batch) equals the levioSAM maps data/lifts/*.lft
    (test/src/lifting_test.cpp:57-141), identity on contig 0."""
import os

import numpy as np
import pytest

from moni_align_amd import capi
from tests import sdsl_reader as sr

D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_data")
LDX = os.path.join(D, "Chr21.10.ldx")
LFT = ["HG00096_H1_21", "HG00096_H2_21", "HG00097_H1_21", "HG00097_H2_21", "HG00099_H1_21", "HG00099_H2_21", "HG00100_H1_21", "HG00100_H2_21"]


def test_fixture_ldx_round_trips_byte_for_byte(tmp_path):
    info = capi.ldx_info(LDX)
    assert info == {"n_seq": 9, "u": 420375141, "w": 0, "has_w": False}
    out = str(tmp_path / "old.ldx")
    capi.ldx_rewrite(LDX, out, False)
    assert open(out, "rb").read() == open(LDX, "rb").read()
    new = str(tmp_path / "new.ldx")
    capi.ldx_rewrite(LDX, new, True)
    assert os.path.getsize(new) == os.path.getsize(LDX) + 8
    assert capi.ldx_info(new) == {"n_seq": 9, "u": 420375141, "w": 10, "has_w": True}
    back = str(tmp_path / "back.ldx")
    capi.ldx_rewrite(new, back, False)
    assert open(back, "rb").read() == open(LDX, "rb").read()


def test_truncated_or_foreign_file_is_rejected(tmp_path):
    bad = str(tmp_path / "bad.ldx")
    open(bad, "wb").write(open(LDX, "rb").read()[:5000])
    with pytest.raises(RuntimeError):
        capi.ldx_info(bad)
    open(bad, "wb").write(b"not an index" * 10)
    with pytest.raises(RuntimeError):
        capi.ldx_info(bad)


@pytest.mark.gpu
def test_gpu_lift_of_all_eight_haplotypes_equals_leviosam_maps():
    d = sr.read_ldx_old_layout(open(LDX, "rb").read())
    starts = d["starts"].ones.astype(np.uint64)
    rng = np.random.default_rng(11)
    p0 = rng.integers(0, int(starts[1]) - 10, size=5000).astype(np.uint64)
    assert np.array_equal(capi.ldx_lift_batch(LDX, p0), p0)                       # contig 0: identity (lifting_test.cpp:117-121)
    for j, name in enumerate(LFT, start=1):
        c = sr.Cursor(open(os.path.join(D, name + ".lft"), "rb").read())
        assert c.u64() == 1
        M = sr.Lift(c)
        zeros = M.dele.size - M.dele.m
        ps = np.unique(rng.integers(0, zeros, size=3000)).astype(np.uint64)
        want = np.array([M.lift_pos(int(p)) for p in ps], dtype=np.uint64)
        assert np.array_equal(capi.ldx_lift_batch(LDX, ps + starts[j]), want), name
