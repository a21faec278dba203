"""Parity at the sizes BASELINE.json names (-m gpu; the indexes are built on the GPU by index_build exactly as bench.py builds
them and cached under /tmp for the other tests of the session):

  * configs[2]: mouse-chr19-scale base (61,420,004 bp) + 12 haplotypes, ref+VCF -H12 style (haplotypes lift onto the reference
    contig), n = 798 M, r = 46 M (run indices beyond 2^24): seeds (MEMs, halves, occurrence lists) and SAM text of 50,000 x 150 bp
    reads identical to the CPU oracle's.  (Runs of 4095 and more, whose offsets leave the 12-bit fast rows, do not occur in an
    i.i.d. genome at any size: tests/test_gpu_seed.py::test_long_runs_take_the_general_path builds one on purpose.)
  * configs[4]-shaped: chr21-scale base (46,709,983 bp) + 20 haplotypes, 250 bp reads: 10,000 reads.
  * the mouse-scale index with 5 % interspersed repeats (seed 1919): MEMs with many occurrences, reads beyond the
    staged kernels' capacities (taken by the general kernel / the host pipeline), per-genome caps.
Independent properties of every SAM record are checked in tests/test_sam_properties.py on the same outputs."""
import os
import time

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]

CACHE = os.environ.get("MONI_TEST_CACHE", "/tmp/moni_bench_cache")


_MEMO = {}


def build_or_load(base_len, haps, repeats=0.0):
    key = (base_len, haps, repeats)
    if key not in _MEMO:
        _MEMO.clear()                      # one full-size index in host memory at a time
        _MEMO[key] = _build_or_load(base_len, haps, repeats)
    return _MEMO[key]


def _build_or_load(base_len, haps, repeats=0.0):
    import torch
    from moni_align_amd import index_build, synth
    pg = synth.make_pangenome(base_len, haps, seed=19, var_seed=12, repeat_frac=repeats)
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, "idx_%d_%d_lifted_%g.mfi" % (base_len, haps, repeats))      # the name bench.py uses
    if os.path.exists(path):
        return pg, index_build.FlatIndex.load(path)
    t0 = time.time()
    fi = index_build.build_from_pangenome(pg, device="cuda:0")
    torch.cuda.empty_cache()
    print("built n=%d r=%d in %.0fs" % (fi.n, fi.r, time.time() - t0), flush=True)
    return pg, fi


def first_diff(a: bytes, b: bytes):
    la, lb = a.split(b"\n"), b.split(b"\n")
    for k, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return k, x.decode()[:400], y.decode()[:400]
    return min(len(la), len(lb)), "<%d records>" % len(la), "<%d records>" % len(lb)


def check(pg, fi, n_reads, L, seed, threads=16, seeds_too=True):
    from moni_align_amd import capi, synth
    from oracle import orc
    from tests.parity import assert_seeds_equal
    reads = synth.make_reads(pg, n_reads, L, seed=seed)
    offs = np.arange(0, (n_reads + 1) * L, L, dtype=np.uint64)
    names, noff = orc.make_names(n_reads)
    quals = np.full(n_reads * L, ord("I"), dtype=np.uint8)
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        o = orc.OracleIndex(fi=fi)
        if seeds_too:
            ctx.upload(reads.reshape(-1), offs)
            ctx.seed_run(25, True, 1000)
            assert_seeds_equal(ctx.seed_fetch(), o.seed_batch(reads.reshape(-1), offs, 25, True, 1000, threads=threads))
        got, st = ctx.align_batch(reads.reshape(-1), offs, names, noff, quals, host_threads=threads)
        want, wc = orc.align_batch(o, reads.reshape(-1), offs, names, noff, quals, threads=threads)
        if got != want:
            raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))
        assert st["aligned"] == wc["aligned"]
        # moni_align_run: reads resident in HBM, lines put in read order by gather_lines_kernel, one transfer per sub-batch
        ctx.upload(reads.reshape(-1), offs)
        got_run, st_run = ctx.align_run(names, noff, quals, host_threads=threads)
        if got_run != want:
            raise AssertionError("moni_align_run SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got_run, want))
        assert st_run["aligned"] == wc["aligned"]
        # several sub-batches: seeding slice by slice beside the align kernels of the previous slice, per-MEM buffers that have to grow
        # mid-batch (small estimates), the streaming entry point
        os.environ["MONI_ALIGN_SUB"] = str(max(1000, n_reads // 7)); os.environ["MONI_SEED_EST"] = "0.5,2"
        try:
            got_p, st_p = ctx.align_batch(reads.reshape(-1), offs, names, noff, quals, host_threads=threads, stream=True)
        finally:
            del os.environ["MONI_ALIGN_SUB"], os.environ["MONI_SEED_EST"]
        if got_p != want:
            raise AssertionError("pipelined moni_align_stream SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got_p, want))
        return got, st
    finally:
        ctx.close()
        idx.close()


def test_configs2_mouse_scale_lifted_index_50k_reads():
    pg, fi = build_or_load(61420004, 12)
    assert fi.n > 790_000_000 and fi.r > 40_000_000 and fi.lifts is not None
    assert fi.r > (1 << 24)                                     # run indices need the high part of the 32-bit `dest` fields (image.hpp packing)
    sam, st = check(pg, fi, 50000, 150, seed=150)
    assert st["aligned"] > 49900 and st["handed_back"] == 0
    assert st["kernel_fallback"] < 50                           # the staged kernels take (nearly) every read at this size
    rn = set(l.split(b"\t")[2] for l in sam.split(b"\n") if l)
    assert rn <= {b"chr19", b"*"}                               # every record is lifted onto the reference contig


def test_configs2_paired_end_with_orphan_recovery():
    """pairs on the configs[2]-scale index (40-bit positions, run indices beyond 2^24, lifted haplotypes): st_align's paired loop through the C ABI
    (learn on batches of 512, align them, then the rest) against the oracle's paired path with orphan recovery; fragment model bit-identical"""
    from moni_align_amd import capi, synth
    from oracle import orc
    from tests.test_gpu_pe import gpu_align_all
    pg, fi = build_or_load(61420004, 12)
    N, L = 6000, 150
    mates, _ = synth.make_pairs(pg, N, L, seed=351)
    rng = np.random.default_rng(7)
    for p in range(40, N, 97):                     # pairs that fail jointly: one mate with a substitution every 17 bases (no 25-base MEM) -> orphan recovery
        x = mates[2 * p + (p & 1)]
        x[3::17] = np.where(x[3::17] == ord("A"), ord("C"), ord("A"))
    for p in range(13, N, 211):                    # one mate of noise
        mates[2 * p + 1] = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
    names, noff = synth.make_pair_names(N)
    offs = np.arange(0, (2 * N + 1) * L, L, dtype=np.uint64)
    q = ((np.arange(2 * N * L) % 39) + 34).astype(np.uint8)
    o = orc.OracleIndex(fi=fi)
    o1 = np.arange(0, (N + 1) * L, L, dtype=np.uint64)
    n1 = b"".join(bytes(names[int(noff[2 * p]):int(noff[2 * p + 1])]) for p in range(N)); n2 = b"".join(bytes(names[int(noff[2 * p + 1]):int(noff[2 * p + 2])]) for p in range(N))
    no1 = np.zeros(N + 1, np.uint64); no1[1:] = np.cumsum([int(noff[2 * p + 1] - noff[2 * p]) for p in range(N)])
    no2 = np.zeros(N + 1, np.uint64); no2[1:] = np.cumsum([int(noff[2 * p + 2] - noff[2 * p + 1]) for p in range(N)])
    qq = q.reshape(2 * N, L)
    want, st = orc.align_pe(o, np.ascontiguousarray(mates[0::2]).reshape(-1), o1, np.ascontiguousarray(mates[1::2]).reshape(-1), o1, np.frombuffer(n1, np.uint8), no1,
                            np.frombuffer(n2, np.uint8), no2, np.ascontiguousarray(qq[0::2]).reshape(-1), np.ascontiguousarray(qq[1::2]).reshape(-1), b_size=512, find_orphan=True)
    assert st["orphan_recovered"] > 20
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        got, model, aligned = gpu_align_all(ctx, mates.reshape(-1), offs, names, noff, q, 512, find_orphan=True)
    finally:
        ctx.close()
        idx.close()
    assert model.complete and model.count == st["ins_count"] and model.mean == st["ins_mean"] and model.std_dev == st["ins_std_dev"]
    if got != want:
        raise AssertionError("paired SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))
    assert aligned == st["aligned"] and aligned > 5800
    # -Z (find_chains_secondary) on the same pairs and index: the staged kernels' second track against the oracle's
    want_z, st_z = orc.align_pe(o, np.ascontiguousarray(mates[0::2]).reshape(-1), o1, np.ascontiguousarray(mates[1::2]).reshape(-1), o1, np.frombuffer(n1, np.uint8), no1,
                                np.frombuffer(n2, np.uint8), no2, np.ascontiguousarray(qq[0::2]).reshape(-1), np.ascontiguousarray(qq[1::2]).reshape(-1), b_size=512, find_orphan=True,
                                secondary_chains=True)
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        got_z, model_z, aligned_z = gpu_align_all(ctx, mates.reshape(-1), offs, names, noff, q, 512, find_orphan=True, secondary_chains=1)
    finally:
        ctx.close()
        idx.close()
    assert model_z.count == st_z["ins_count"] and model_z.mean == st_z["ins_mean"] and model_z.std_dev == st_z["ins_std_dev"]
    if got_z != want_z:
        raise AssertionError("paired SAM with -Z differs at record %d:\n got: %s\nwant: %s" % first_diff(got_z, want_z))
    assert aligned_z == st_z["aligned"]


def test_configs4_shaped_chr21_scale_20_haplotypes_250bp():
    pg, fi = build_or_load(46709983, 20)
    assert fi.n > 970_000_000 and len(fi.names) == 21
    sam, st = check(pg, fi, 10000, 250, seed=250)
    assert st["aligned"] > 9900


def test_mouse_scale_with_interspersed_repeats():
    pg, fi = build_or_load(61420004, 12, repeats=0.05)
    sam, st = check(pg, fi, 20000, 150, seed=1919)
    # repeat copies: seeds with many occurrences, more chains than the staged kernels' small instance holds
    print("repeat-rich index: %d of 20000 reads to the general kernel, %d to the host pipeline" % (st["kernel_fallback"], st["handed_back"]), flush=True)
    assert st["aligned"] > 19000
