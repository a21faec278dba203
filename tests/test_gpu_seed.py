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
