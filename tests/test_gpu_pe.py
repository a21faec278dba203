"""Paired-end on the GPU (moni_pe_learn_batch / moni_pe_align_batch: seeding kernels + pe_align_kernel + the host finishing) against the
oracle's restatement of the reference's paired path without orphan recovery (oracle/align_pe.hpp): the learnt fragment model must be
bit-identical and the SAM text byte-identical, batch order as in st_align (learn on the first batches, align them, then the rest)."""
import numpy as np
import pytest

from moni_align_amd import capi, index_build, synth
from oracle import orc
from tests.test_host_sim_pe import first_diff, hard_pairs, interleave, oracle_pe
from tests.test_oracle_pe import make_pairs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case():
    pg = synth.make_pangenome(80000, 3, site_spacing=800)
    fi = index_build.build_from_pangenome(pg, device="cpu")
    return pg, fi, orc.OracleIndex(fi=fi)


def gpu_align_all(ctx, seq, offs, names, noff, q, b_size, find_orphan=False, secondary_chains=0):
    """st_align's paired loop (align_reads_dispatcher.hpp:356-389) over the C ABI"""
    n = (len(offs) - 1) // 2
    model = capi.PeModelC()
    at, out, learnt = 0, [], []
    def cut(lo, hi):
        sl = slice(2 * lo, 2 * hi + 1)
        return (seq[int(offs[2 * lo]):int(offs[2 * hi])], offs[sl] - offs[2 * lo], names[int(noff[2 * lo]):int(noff[2 * hi])], noff[sl] - noff[2 * lo],
                q[int(offs[2 * lo]):int(offs[2 * hi])])
    while at < n:
        hi = min(n, at + b_size)
        s, o, _, _, _ = cut(at, hi)
        learnt.append((at, hi))
        at = hi
        ctx.pe_learn(s, o, model, secondary_chains=secondary_chains)
        if model.complete:
            break
    aligned = 0
    rest = [(lo, min(n, lo + b_size)) for lo in range(at, n, b_size)]
    for lo, hi in learnt + rest:
        sam, st = ctx.pe_align(*cut(lo, hi), model, host_threads=4, find_orphan=int(find_orphan), secondary_chains=secondary_chains)
        out.append(sam); aligned += st["aligned"]
    return b"".join(out), model, aligned


def on_gpu(fi, seq, offs, names, noff, q, b_size, find_orphan=False):
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        return gpu_align_all(ctx, seq, offs, names, noff, q, b_size, find_orphan)
    finally:
        ctx.close()
        idx.close()


def test_pe_matches_oracle(case):
    pg, fi, o = case
    m1, m2, _ = make_pairs(pg, 1300)
    want, st = oracle_pe(o, m1, m2, b_size=512)
    seq, offs, names, noff, q = interleave(m1, m2)
    got, model, aligned = on_gpu(fi, seq, offs, names, noff, q, 512)
    assert model.complete and model.count == st["ins_count"] and model.mean == st["ins_mean"] and model.std_dev == st["ins_std_dev"]
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))
    assert aligned == st["aligned"] and aligned > 1200


def test_pe_hard_cases(case):
    pg, fi, o = case
    m1, m2 = hard_pairs(pg)
    want, st = oracle_pe(o, m1, m2, slash=False, b_size=4096)
    seq, offs, names, noff, q = interleave(m1, m2, slash=False)
    got, model, aligned = on_gpu(fi, seq, offs, names, noff, q, 4096)
    assert not model.complete and model.count == st["ins_count"] and model.mean == st["ins_mean"] and model.std_dev == st["ins_std_dev"]
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))


def test_pe_250bp_lifted_index(case):
    """longer mates, more indels, a second index shape (5 haplotypes)"""
    pg = synth.make_pangenome(60000, 5, site_spacing=400)
    fi = index_build.build_from_pangenome(pg, device="cpu")
    o = orc.OracleIndex(fi=fi)
    m1, m2, _ = make_pairs(pg, 700, L=250, mean=700.0, sd=60.0, seed=77)
    want, st = oracle_pe(o, m1, m2, b_size=512)
    seq, offs, names, noff, q = interleave(m1, m2)
    got, model, aligned = on_gpu(fi, seq, offs, names, noff, q, 512)
    assert model.count == st["ins_count"] and model.mean == st["ins_mean"] and model.std_dev == st["ins_std_dev"]
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))


def test_pe_host_pipeline_forced(case, monkeypatch):
    """every third pair is treated as beyond pe_align_kernel's capacities (MONI_PE_FORCE_BIG): it goes through the host pipeline for pairs
    (pe_big.cpp: the same state machine with large capacities, DP batches on the GPU) and must come out the same"""
    pg, fi, o = case
    m1, m2 = hard_pairs(pg, n=400, seed=31)
    want, st = oracle_pe(o, m1, m2, slash=False, b_size=4096)
    seq, offs, names, noff, q = interleave(m1, m2, slash=False)
    monkeypatch.setenv("MONI_PE_FORCE_BIG", "3")
    got, model, aligned = on_gpu(fi, seq, offs, names, noff, q, 4096)
    assert model.count == st["ins_count"] and model.mean == st["ins_mean"] and model.std_dev == st["ins_std_dev"]
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))


def test_pe_exact_repeats_overflow_the_kernel():
    """a 600-base segment copied 60 times: every MEM inside it has hundreds of occurrences, the pairs exceed the kernel's 512 anchors and
    are taken by the host pipeline for pairs (stats: handed_back)"""
    rng = np.random.default_rng(11)
    base = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=60000)].copy()
    seg = base[1000:1600].copy()
    for k in range(60):
        base[2000 + k * 900:2600 + k * 900] = seg
    pg = synth.make_pangenome(60000, 4, site_spacing=2500, base=base)
    fi = index_build.build_from_pangenome(pg, device="cpu")
    o = orc.OracleIndex(fi=fi)
    m1, m2, _ = make_pairs(pg, 300, L=100, mean=350, sd=30, seed=5)
    want, st = oracle_pe(o, m1, m2, b_size=1024)
    seq, offs, names, noff, q = interleave(m1, m2)
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        model = ctx.pe_learn(seq, offs)
        got, gst = ctx.pe_align(seq, offs, names, noff, q, model, host_threads=4, find_orphan=0)
    finally:
        ctx.close(); idx.close()
    assert model.count == st["ins_count"] and model.mean == st["ins_mean"] and model.std_dev == st["ins_std_dev"]
    assert gst["handed_back"] > 50
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))


def test_pe_chunking_and_wave_shape_are_invisible(case, monkeypatch):
    """many small chunks (the double-buffered records / pools are reused every second chunk, the host finishing of one chunk runs beside
    the next chunk's kernel) and another number of pairs per wave must give the same bytes as one chunk - and the oracle's"""
    pg, fi, o = case
    m1, m2, _ = make_pairs(pg, 6000, seed=41)
    want, st = oracle_pe(o, m1, m2, b_size=512)
    seq, offs, names, noff, q = interleave(m1, m2)
    a, model_a, _ = on_gpu(fi, seq, offs, names, noff, q, 512)
    monkeypatch.setenv("MONI_PE_CHUNK", "700")
    monkeypatch.setenv("MONI_PE_NL", "5")
    b, model_b, _ = on_gpu(fi, seq, offs, names, noff, q, 512)
    assert model_a.mean == st["ins_mean"] and model_b.mean == st["ins_mean"] and model_b.std_dev == st["ins_std_dev"]
    assert a == b
    if a != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(a, want))


def test_pe_orphan_recovery(case, monkeypatch):
    """pairs that chain but fail jointly (one mate without a 25-base MEM): with find_orphan the mate is searched by local alignment (klib's
    ksw_align, here pe_sw_local: one lane per request) in the window the model predicts - in the kernel, and in the host pipeline for pairs when
    every third pair is forced through it; SAM identical to the oracle's orphan_recovery"""
    from tests.test_host_sim_pe import seedless_pairs
    pg, fi, o = case
    m1, m2 = seedless_pairs(pg)
    want, st = oracle_pe(o, m1, m2, b_size=4096, find_orphan=True)
    assert st["orphan_recovered"] > 20
    seq, offs, names, noff, q = interleave(m1, m2)
    got, model, aligned = on_gpu(fi, seq, offs, names, noff, q, 4096, find_orphan=True)
    assert model.mean == st["ins_mean"] and model.std_dev == st["ins_std_dev"]
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))
    assert aligned == st["aligned"]
    monkeypatch.setenv("MONI_PE_SW_WAVE", "0")           # the one-lane form of the local alignment
    got1, _, _ = on_gpu(fi, seq, offs, names, noff, q, 4096, find_orphan=True)
    assert got1 == want
    monkeypatch.delenv("MONI_PE_SW_WAVE")
    monkeypatch.setenv("MONI_PE_FORCE_BIG", "3")
    got2, _, _ = on_gpu(fi, seq, offs, names, noff, q, 4096, find_orphan=True)
    assert got2 == want


@pytest.mark.parametrize("filter_dir", [1, 0])
def test_pe_report_mems(case, filter_dir):
    """-m for pairs (moni_pe_report_mems_batch): one secondary record per occurrence of every MEM the direction and frequency filters leave,
    in the order of the reference's four find_mems calls, against the oracle (aligner_ksw2.hpp:1118-1180)"""
    pg, fi, o = case
    m1, m2, _ = make_pairs(pg, 300)
    h1, h2 = hard_pairs(pg)
    m1, m2 = list(m1) + list(h1), list(m2) + list(h2)
    n = len(m1)
    o1 = np.zeros(n + 1, np.uint64); o1[1:] = np.cumsum([len(x) for x in m1])
    o2 = np.zeros(n + 1, np.uint64); o2[1:] = np.cumsum([len(x) for x in m2])
    nm1 = [b"p%d/1" % i for i in range(n)]; nm2 = [b"p%d/2" % i for i in range(n)]
    no1 = np.zeros(n + 1, np.uint64); no1[1:] = np.cumsum([len(x) for x in nm1])
    no2 = np.zeros(n + 1, np.uint64); no2[1:] = np.cumsum([len(x) for x in nm2])
    q1 = ((np.arange(int(o1[-1])) % 40) + 33).astype(np.uint8); q2 = ((np.arange(int(o2[-1])) % 37) + 35).astype(np.uint8)
    want, _ = orc.align_pe(o, np.concatenate(m1), o1, np.concatenate(m2), o2, np.frombuffer(b"".join(nm1), np.uint8), no1, np.frombuffer(b"".join(nm2), np.uint8), no2,
                           q1, q2, b_size=512, report_mems=True, filter_dir=bool(filter_dir))
    assert want.count(b"\t272\t") > 100 and want.count(b"\t256\t") > 100
    seq, offs, names, noff, _ = interleave(m1, m2)
    q = np.concatenate([x for p in range(n) for x in (q1[int(o1[p]):int(o1[p + 1])], q2[int(o2[p]):int(o2[p + 1])])])
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        got = ctx.pe_report_mems(seq, offs, names, noff, q, filter_dir=filter_dir)
    finally:
        ctx.close()
        idx.close()
    if got != want:
        raise AssertionError("MEM records differ at record %d:\n got: %s\nwant: %s" % first_diff(got, want))


@pytest.mark.parametrize("filter_dir", [1, 0])
def test_pe_csv_statistics(case, filter_dir):
    """-c for pairs (moni_pe_align_csv_batch; aligner_ksw2.hpp:888-918, 1030-1031, 1066-1075, 1115-1118, 1354-1358; csv.hpp:55-67): one line per pair under
    mate 1's name without its slash suffix - MEMs of both mates and their halves, their occurrences, the extreme frequencies and per-genome counts, the
    occurrences the direction and frequency filters drop, the chains check_paired_left_MEM skips - and the unchanged SAM records: both equal the oracle's."""
    pg, fi, o = case
    a1, a2 = hard_pairs(pg, n=260, seed=43)
    b1, b2, _ = make_pairs(pg, 340)
    m1, m2 = list(a1) + list(b1), list(a2) + list(b2)
    n = len(m1)
    want_sam, st = oracle_pe(o, m1, m2, b_size=n, filter_dir=bool(filter_dir))
    want_csv, _ = oracle_pe(o, m1, m2, b_size=n, filter_dir=bool(filter_dir), csv=True)
    seq, offs, names, noff, q = interleave(m1, m2)
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        model = capi.PeModelC()
        ctx.pe_learn(seq, offs, model, filter_dir=filter_dir)
        sam, csv, gst = ctx.pe_align_csv(seq, offs, names, noff, q, model, host_threads=4, filter_dir=filter_dir, find_orphan=0)
    finally:
        ctx.close()
        idx.close()
    assert model.count == st["ins_count"] and model.mean == st["ins_mean"]
    if sam != want_sam:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(sam, want_sam))
    if csv != want_csv:
        raise AssertionError("CSV differs at line %d:\n got: %s\nwant: %s" % first_diff(csv, want_csv))
    cols = np.array([[float(x) for x in ln.split(b",")[1:]] for ln in csv.split(b"\n") if ln])
    assert cols.shape == (n, 8) and cols[:, 0].max() > 2 and cols[:, 7].sum() > 0          # chains were skipped somewhere
    assert csv.split(b"\n")[0].startswith(b"p0,")                                            # remove_slash_mate


def test_pe_stream_variant_and_small_chunks(case, monkeypatch):
    """moni_pe_align_stream (text in the context's buffer, kept across calls) = moni_pe_align_batch; many chunks in flight (every chunk owns its
    records and pools, the hand-over kernels run on several streams)"""
    pg, fi, o = case
    m1, m2, _ = make_pairs(pg, 1500)
    seq, offs, names, noff, q = interleave(m1, m2)
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        model = capi.PeModelC()
        ctx.pe_learn(seq, offs, model)
        a, sa = ctx.pe_align(seq, offs, names, noff, q, model, host_threads=4, find_orphan=1)
        monkeypatch.setenv("MONI_PE_CHUNK", "64")
        for _ in range(2):
            b, sb = ctx.pe_align(seq, offs, names, noff, q, model, host_threads=3, find_orphan=1, stream=True)
            assert a == b and sa["aligned"] == sb["aligned"]
        # moni_pe_align_run: the mates made resident by moni_reads_upload (what bench.py --paired times)
        monkeypatch.delenv("MONI_PE_CHUNK")
        ctx.upload(seq, offs)
        r, sr = ctx.pe_align_run(names, noff, q, model, host_threads=4, find_orphan=1)
        assert r == a and sr["aligned"] == sa["aligned"]
        r2, _ = ctx.pe_align_run(names, noff, None, model, host_threads=4, find_orphan=1)          # without qualities
        assert r2.count(b"\n") == a.count(b"\n") and r2 != a and b"\t*\tAS:i:" in r2
        monkeypatch.setenv("MONI_PE_CHUNK", "64")
        half = 2 * 700
        c_, _ = ctx.pe_align(seq[:int(offs[half])], offs[:half + 1], names[:int(noff[half])], noff[:half + 1], q[:int(offs[half])], model, host_threads=4, find_orphan=1, stream=True)
        assert c_ == b"".join(a.split(b"\n")[i] + b"\n" for i in range(half))
    finally:
        ctx.close()
        idx.close()


def test_pe_lines_written_on_the_gpu_equal_the_host_finishing(case, monkeypatch):
    """pe_lines_kernel (two lines per pair written and ordered on the GPU) against pe_host.hpp's finishing of the same records
    (MONI_PE_HOST_FORMAT=1), over proper pairs, hard cases (noise mates, improper pairs, one mate under its minimum score), orphan recovery,
    names without /1 /2 and input without qualities"""
    pg, fi, o = case
    m1, m2, _ = make_pairs(pg, 1200)
    h1, h2 = hard_pairs(pg)
    from tests.test_host_sim_pe import seedless_pairs
    s1, s2 = seedless_pairs(pg)
    m1, m2 = list(m1) + list(h1) + list(s1), list(m2) + list(h2) + list(s2)
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        for slash, with_q in ((True, True), (False, False)):
            seq, offs, names, noff, q = interleave(m1, m2, slash=slash)
            q = ((np.arange(len(seq)) % 41) + 33).astype(np.uint8) if with_q else None
            model = capi.PeModelC()
            ctx.pe_learn(seq[:int(offs[2400])], offs[:2401], model)
            monkeypatch.delenv("MONI_PE_HOST_FORMAT", raising=False)
            a, sa = ctx.pe_align(seq, offs, names, noff, q, model, host_threads=4, find_orphan=1)
            monkeypatch.setenv("MONI_PE_HOST_FORMAT", "1")
            b, sb = ctx.pe_align(seq, offs, names, noff, q, model, host_threads=4, find_orphan=1)
            monkeypatch.delenv("MONI_PE_HOST_FORMAT", raising=False)
            if a != b:
                raise AssertionError("lines differ at record %d:\n gpu: %s\nhost: %s" % first_diff(a, b))
            assert sa["aligned"] == sb["aligned"] and a.count(b"\n") == 2 * len(m1)
    finally:
        ctx.close()
        idx.close()


def test_pe_orphan_loop_over_several_waves(case, monkeypatch):
    """pe_orphan_kernel: a pair that needs orphan recovery is parked where its chain loop begins, the chains are scored by `nsplit` waves (records per
    chain), the loop is replayed from the records in chain order.  Same SAM as one wave doing it all (MONI_PE_NSPLIT=0), as the oracle, and as the runs
    where only the first few listed pairs may park (the rest: one wave each) or every chain is its own part"""
    from tests.test_host_sim_pe import seedless_pairs, oracle_pe_orphan
    pg, fi, o = case
    m1, m2 = seedless_pairs(pg)
    want, st = oracle_pe_orphan(o, m1, m2, b_size=4096)
    assert st["orphan_recovered"] > 20
    seq, offs, names, noff, q = interleave(m1, m2)
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        model = capi.PeModelC()
        model.mean, model.std_dev, model.complete = st["ins_mean"], st["ins_std_dev"], 1
        for env in ({}, {"MONI_PE_NSPLIT": "0"}, {"MONI_PE_NSPLIT": "3", "MONI_PE_OCAP": "5"}, {"MONI_PE_NSPLIT": "64", "MONI_PE_K1_WAVES": "64"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            got, sg = ctx.pe_align(seq, offs, names, noff, q, model, host_threads=4, find_orphan=1)
            for k in env:
                monkeypatch.delenv(k)
            if got != want:
                raise AssertionError("%s: SAM differs at record %d:\n got: %s\nwant: %s" % ((env,) + first_diff(got, want)))
            assert sg["aligned"] == st["aligned"] and sg["kernel_fallback"] > 20
    finally:
        ctx.close()
        idx.close()


def test_pe_secondary_chains(case):
    """-Z (moni_pe_params_t::secondary_chains): find_chains_secondary in learn_fragment_model and in the alignment, st_align's batch order, through the staged
    kernels (af_chain's second track; the pairs they hand over take pe_core.h's).  Against the oracle; and the option must change the output"""
    pg, fi, o = case
    m1, m2, _ = make_pairs(pg, 1300)
    h1, h2 = hard_pairs(pg)
    m1, m2 = list(m1) + list(h1), list(m2) + list(h2)
    want, st = oracle_pe(o, m1, m2, b_size=512, find_orphan=True, secondary_chains=True)
    plain, _ = oracle_pe(o, m1, m2, b_size=512, find_orphan=True)
    assert sum(x != y for x, y in zip(want.split(b"\n"), plain.split(b"\n"))) > 100
    seq, offs, names, noff, q = interleave(m1, m2)
    n = len(m1)
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        model = capi.PeModelC()
        at, learnt = 0, []
        while at < n and not model.complete:
            hi = min(n, at + 512)
            ctx.pe_learn(seq[int(offs[2 * at]):int(offs[2 * hi])], offs[2 * at:2 * hi + 1] - offs[2 * at], model, secondary_chains=1)
            learnt.append((at, hi)); at = hi
        assert model.count == st["ins_count"] and model.mean == st["ins_mean"] and model.std_dev == st["ins_std_dev"]
        out = []
        for lo, hi in learnt + [(x, min(n, x + 512)) for x in range(at, n, 512)]:
            sl = slice(2 * lo, 2 * hi + 1)
            sam, _ = ctx.pe_align(seq[int(offs[2 * lo]):int(offs[2 * hi])], offs[sl] - offs[2 * lo], names[int(noff[2 * lo]):int(noff[2 * hi])], noff[sl] - noff[2 * lo],
                                  q[int(offs[2 * lo]):int(offs[2 * hi])], model, host_threads=4, find_orphan=1, secondary_chains=1)
            out.append(sam)
        got = b"".join(out)
    finally:
        ctx.close()
        idx.close()
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))
