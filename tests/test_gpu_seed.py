"""Parity of the HIP seeding path (through the C ABI of libmoni_hip.so) with the CPU oracle on a real
MI355X: bit-exact MS pointers, MEMs, halves, occurrence lists, filter counts and work counters."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def ragged(reads_list):
    offs = np.zeros(len(reads_list) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads_list])
    seq = np.concatenate(reads_list) if reads_list else np.zeros(0, np.uint8)
    return seq, offs


@pytest.fixture(scope="module")
def gpu_pair(medium_case):
    from moni_align_amd import capi
    from oracle import orc
    idx = capi.Index(fi=medium_case.fi)
    ctx = capi.Ctx(idx)
    yield orc.OracleIndex(medium_case.path), ctx
    ctx.close()
    idx.close()


def run_both(pair, seq, offs, min_len=25, filter_seeds=True, n_seeds_thr=1000):
    o, ctx = pair
    want = o.seed_batch(seq, offs, min_len, filter_seeds, n_seeds_thr, threads=4)
    ctx.upload(seq, offs)
    ctx.seed_run(min_len, filter_seeds, n_seeds_thr)
    got = ctx.seed_fetch()
    got["counters"] = ctx.counters()
    return got, want


def test_ms_pointers(medium_case, gpu_pair):
    o, ctx = gpu_pair
    reads = medium_case.synth.make_reads(medium_case.pg, 500, 150, seed=150)
    seq, offs = ragged(list(reads))
    ptr = ctx.ms_query_batch(seq, offs)
    rc = medium_case.synth.revcomp(reads)
    for i in range(500):
        assert np.array_equal(ptr[300 * i:300 * i + 150], o.ms_query(reads[i].tobytes())), i
        assert np.array_equal(ptr[300 * i + 150:300 * (i + 1)], o.ms_query(rc[i].tobytes())), i


def test_seeds_150bp(medium_case, gpu_pair):
    from tests.parity import assert_seeds_equal
    reads = medium_case.synth.make_reads(medium_case.pg, 20000, 150, seed=151)
    seq, offs = ragged(list(reads))
    got, want = run_both(gpu_pair, seq, offs)
    assert_seeds_equal(got, want)
    assert np.array_equal(got["counters"], want["counters"])


def test_seeds_ragged_edge_cases(medium_case, gpu_pair):
    from tests.parity import assert_seeds_equal
    rng = np.random.default_rng(5)
    base = medium_case.synth.make_reads(medium_case.pg, 3000, 250, seed=9, sub_rate=0.03, indel_rate=0.003)
    reads = [r[: int(rng.integers(1, 251))].copy() for r in base]
    reads.append(np.zeros(0, np.uint8))
    reads.append(np.frombuffer(b"N" * 60, dtype=np.uint8))
    reads.append(np.frombuffer(b"acgtacgtacgtacgtacgtacgtacgtacgtacgt", dtype=np.uint8))
    x = base[0].copy(); x[40:45] = ord("N"); reads.append(x)
    reads.append(np.frombuffer(bytes(medium_case.fi.text[100:400]), dtype=np.uint8))
    seq, offs = ragged(reads)
    got, want = run_both(gpu_pair, seq, offs)
    assert_seeds_equal(got, want)
    assert np.array_equal(got["counters"], want["counters"])


def test_one_long_read_among_short_ones(medium_case, gpu_pair):
    """a 20 kb read in a batch of 150 bp reads: the per-step workspaces (packed patterns, MS pointers) are laid out per block of 32 reads
    over the block's own longest read (seed_core.h), so the long read costs 64 x 20 k words in its own block instead of multiplying the
    whole batch's stride; seeds and MS pointers still equal the oracle's"""
    from tests.parity import assert_seeds_equal
    rng = np.random.default_rng(11)
    reads = list(medium_case.synth.make_reads(medium_case.pg, 5000, 150, seed=12))
    text = np.frombuffer(bytes(medium_case.fi.text[3000:23000]), dtype=np.uint8).copy()
    k = rng.integers(0, len(text), size=200)
    text[k] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=len(k))]
    reads.insert(1777, text)
    reads.insert(40, text[:3000].copy())
    seq, offs = ragged(reads)
    got, want = run_both(gpu_pair, seq, offs)
    assert_seeds_equal(got, want)
    assert np.array_equal(got["counters"], want["counters"])
    o, ctx = gpu_pair
    part = reads[1760:1800]                                   # holds the 20 kb read: the host-buffer form de-interleaves the block layout
    pseq, poffs = ragged(part)
    ptr = ctx.ms_query_batch(pseq, poffs)
    for i, r in enumerate(part):
        a, m = 2 * int(poffs[i]), len(r)
        assert np.array_equal(ptr[a:a + m], o.ms_query(r.tobytes())), i
        assert np.array_equal(ptr[a + m:a + 2 * m], o.ms_query(medium_case.synth.revcomp(r[None, :])[0].tobytes())), i


def test_seeds_filter(medium_case, gpu_pair):
    from tests.parity import assert_seeds_equal
    reads = medium_case.synth.make_reads(medium_case.pg, 2000, 150, seed=21, sub_rate=0.005)
    seq, offs = ragged(list(reads))
    for thr in (1, 3):
        got, want = run_both(gpu_pair, seq, offs, n_seeds_thr=thr)
        assert_seeds_equal(got, want)
    got, want = run_both(gpu_pair, seq, offs, filter_seeds=False)
    assert_seeds_equal(got, want)
    got, want = run_both(gpu_pair, seq, offs, min_len=12)
    assert_seeds_equal(got, want)


def test_empty_batch(gpu_pair):
    o, ctx = gpu_pair
    ctx.upload(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    ctx.seed_run()
    got = ctx.seed_fetch()
    assert len(got["mems"]) == 0 and len(got["occs"]) == 0


def test_phi_batch(medium_case, gpu_pair):
    o, ctx = gpu_pair
    fi = medium_case.fi
    rng = np.random.default_rng(1)
    pos = rng.integers(0, fi.n, size=5000).astype(np.uint64)
    first = (int(fi.ssa[0]) + 1) % fi.n
    last = (int(fi.esa[-1]) + 1) % fi.n
    pos = pos[(pos != first) & (pos != last)]
    for inv in (False, True):
        a, b = ctx.phi_lcp_batch(pos, inv)
        for k in range(0, len(pos), 50):
            assert (int(a[k]), int(b[k])) == o.phi_lcp(int(pos[k]), inv)


def test_round_trip_properties_large(gpu_pair, medium_case):
    """Size-independent checks on a larger batch: every MEM occurrence spells the MEM, counters add up."""
    o, ctx = gpu_pair
    n_reads, L = 100000, 150
    reads = medium_case.synth.make_reads(medium_case.pg, n_reads, L, seed=77)
    seq, offs = ragged(list(reads))
    ctx.upload(seq, offs)
    ctx.seed_run()
    got = ctx.seed_fetch()
    cnt = ctx.counters()
    assert int(cnt[0]) == 2 * L * n_reads
    m = got["mems"]
    assert (m["occ_cnt"] + m["num_filtered"] == m["total_occ"]).all()
    assert int(m["occ_cnt"].astype(np.int64).sum()) == len(got["occs"])
    text = medium_case.fi.text
    rc = medium_case.synth.revcomp(reads)
    sel = np.random.default_rng(0).integers(0, len(m), size=3000)
    for k in sel:
        e = m[k]
        s = (rc if e["mate"] & 2 else reads)[e["read"]][e["idx"]:e["idx"] + e["len"]]
        for occ in got["occs"][int(e["occ_off"]):int(e["occ_off"]) + int(e["occ_cnt"])]:
            assert np.array_equal(text[int(occ):int(occ) + int(e["len"])], s)


def test_long_runs_take_the_general_path():
    """BWT runs of 4095 positions and more, offsets that leave the 12-bit fields of the fast rows, a letter without a hot slot: the
    LF kernel's general path (rows / cr / recs with absolute positions) on the GPU against the oracle."""
    from moni_align_amd import capi
    from oracle import orc
    from tests.parity import assert_seeds_equal
    from tests.test_host_sim import long_run_case, ragged
    fi, reads = long_run_case()
    sq, offs = ragged(reads)
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        o = orc.OracleIndex(fi=fi)
        ctx.upload(sq, offs)
        ctx.seed_run(20, True, 100)
        assert_seeds_equal(ctx.seed_fetch(), o.seed_batch(sq, offs, 20, True, 100))
        ptr = ctx.ms_query_batch(sq, offs)
        for i in range(0, 400, 11):
            assert np.array_equal(ptr[int(offs[i]) * 2:int(offs[i]) * 2 + 150], o.ms_query(reads[i].tobytes()))
    finally:
        ctx.close()
        idx.close()


def test_legacy_ms_lengths_and_report_mems(medium_case):
    """`moni ms` / `moni mems` (pointers + matching-statistics lengths of the forward strand) and the -m report-MEMs SAM mode of
    `moni align` (aligner_ksw2.hpp:346-373) on the GPU against the oracle's restatements."""
    from moni_align_amd import capi
    from oracle import orc
    reads = medium_case.synth.make_reads(medium_case.pg, 600, 150, seed=171, sub_rate=0.02, indel_rate=0.003)
    reads[5, 30:33] = ord("N")
    offs = np.arange(0, 601 * 150, 150, dtype=np.uint64)
    o = orc.OracleIndex(medium_case.path)
    idx = capi.Index(fi=medium_case.fi)
    ctx = capi.Ctx(idx)
    try:
        ptr, ln = ctx.ms_lengths_batch(reads.reshape(-1), offs)
        for i in range(0, 600, 13):
            wp, wl = o.ms_lengths(reads[i].tobytes())
            assert np.array_equal(ptr[150 * i:150 * (i + 1)], wp) and np.array_equal(ln[150 * i:150 * (i + 1)], wl)
        names, noff = orc.make_names(600)
        for quals in (None, np.full(reads.size, ord("F"), dtype=np.uint8)):
            got = ctx.report_mems_batch(reads.reshape(-1), offs, names, noff, quals)
            want = orc.report_mems_batch(o, reads.reshape(-1), offs, names, noff, quals)
            assert got == want and got.count(b"\n") > 600
    finally:
        ctx.close()
        idx.close()


def test_positions_beyond_2_to_32():
    """The 40-bit packing of the device image (image.hpp: rows, recs, fast rows with their high sample bytes; seed_core.h: `hi << 32`):
    an r-index whose BWT is longer than 2^32 and whose SA samples exceed 2^32.  No such text can be built here (a 4 G suffix array),
    but ms_pointers::_query (moni.hpp:568-624) is mechanical over the run-length BWT, the thresholds and the samples, so a real
    index of a 2 Mbp text is stretched: every run 2560 times as long (F, run starts and thresholds scale with it, single-position
    runs stay below the 12-bit limit of the fast rows, longer ones take the general path), samples scaled past 2^32.  The kernel's
    pointers must equal the oracle's on the same arrays."""
    import dataclasses
    from moni_align_amd import capi, index_build, synth
    from oracle import orc
    pg = synth.make_pangenome(2_000_000, 0, seed=23)
    fi = index_build.build_from_pangenome(pg, device="cuda:0", lifted=False)
    M = np.uint64(2560)
    n2 = int(fi.n) * int(M)
    assert n2 > (1 << 32)
    big = dataclasses.replace(
        fi, n=n2, F=fi.F * M, starts=fi.starts * M, thr=fi.thr * M,
        ssa=fi.ssa * M + np.uint64(7), esa=fi.esa * M + np.uint64(3),
        text=np.full(n2 - 1, ord("A"), dtype=np.uint8),
        seq_starts=np.array([0, n2 // 2, n2 - 10], dtype=np.uint64), names=["a", "b"])       # (a lift covers fewer than 2^32 columns)
    assert int(big.ssa.max()) > (1 << 32) and int(np.diff(big.starts.astype(np.int64)).min()) == 2560
    rng = np.random.default_rng(4)
    reads = [pg.seqs[0][a:a + 120].copy() for a in rng.integers(0, 1_900_000, size=300)]
    for r in reads[::3]:
        r[int(rng.integers(0, 120))] = ord("ACGT"[int(rng.integers(0, 4))])
    offs = np.arange(0, 301 * 120, 120, dtype=np.uint64)
    sq = np.concatenate(reads)
    idx = capi.Index(fi=big)
    ctx = capi.Ctx(idx)
    try:
        o = orc.OracleIndex(fi=big)
        ptr = ctx.ms_query_batch(sq, offs)
        seen_high = False
        for i in range(300):
            want = o.ms_query(reads[i].tobytes())
            assert np.array_equal(ptr[240 * i:240 * i + 120], want), i
            seen_high = seen_high or bool((want >= (1 << 32)).any())
        assert seen_high
    finally:
        ctx.close()
        idx.close()
