"""Full single-end path on the GPU (moni_align_batch: HIP seeding + HIP ksw2 extension + host stages) against the
oracle's restatement of aligner::align: SAM text byte-identical (CIGAR, POS, MAPQ, AS, NM, MD, ZS, OA, AA, flags)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def first_diff(a: bytes, b: bytes):
    la, lb = a.split(b"\n"), b.split(b"\n")
    for k, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return k, x.decode()[:300], y.decode()[:300]
    return min(len(la), len(lb)), "<%d records>" % len(la), "<%d records>" % len(lb)


@pytest.fixture(scope="module")
def env(medium_case):
    from moni_align_amd import capi
    from oracle import orc
    idx = capi.Index(fi=medium_case.fi)
    ctx = capi.Ctx(idx)
    yield orc.OracleIndex(medium_case.path), ctx
    ctx.close()
    idx.close()


def both(env, reads_list, quals=True):
    from oracle import orc
    o, ctx = env
    offs = np.zeros(len(reads_list) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads_list])
    seq = np.concatenate(reads_list) if reads_list else np.zeros(0, dtype=np.uint8)
    names, noff = orc.make_names(len(reads_list))
    q = np.full(len(seq), ord("I"), dtype=np.uint8) if quals else None
    want, wcnt = orc.align_batch(o, seq, offs, names, noff, q, threads=8)
    got, st = ctx.align_batch(seq, offs, names, noff, q, host_threads=8)
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))
    return wcnt, st


def test_sam_identical_150bp(medium_case, env):
    reads = medium_case.synth.make_reads(medium_case.pg, 20000, 150, seed=150)
    wcnt, st = both(env, list(reads))
    assert st["aligned"] == wcnt["aligned"] > 19000


def test_banded_global_problems_equal_the_full_matrix(medium_case, env, monkeypatch):
    """Chains with overlapping anchors are scored by one global alignment of the whole read (aligner_ksw2.hpp:2984-3015); extensions and gap fills are
    banded the same way where a band of 16 diagonals is proven (bin_tasks_kernel: af_ext_band / af_global_band).  dp_band_kernel computes the band
    of diagonals global_band_kernel has bounded (lower bound from a two-piece diagonal alignment, upper bound from the gap bases a path off the band must
    hold); a problem whose band is wider than 16 diagonals takes the full-matrix kernel.  Substitutions (narrow bands), indels (bands around tlen - qlen,
    wide ones too) and both strands: the SAM text equals the oracle's, and nothing changes when every global problem is forced through the full matrix."""
    reads = list(medium_case.synth.make_reads(medium_case.pg, 6000, 150, seed=161, sub_rate=0.02, indel_rate=0.004)) + \
            list(medium_case.synth.make_reads(medium_case.pg, 3000, 250, seed=162, sub_rate=0.03, indel_rate=0.002)) + \
            list(medium_case.synth.make_reads(medium_case.pg, 3000, 100, seed=163))
    _, st = both(env, reads)
    monkeypatch.setenv("MONI_AF_DBG", "65536")                    # global problems through the full matrix
    _, st_g = both(env, reads)
    monkeypatch.setenv("MONI_AF_DBG", str(65536 + 131072))       # ... and the extensions / gap fills through the tile kernels: no band anywhere
    _, st_full = both(env, reads)
    assert st["aligned"] == st_g["aligned"] == st_full["aligned"] and st["dp_cells"] == st_g["dp_cells"] == st_full["dp_cells"]
    assert st["dp_cells_cut"] < st_g["dp_cells_cut"] < st_full["dp_cells_cut"]          # the banded problems step through fewer cells
    assert st["handed_back"] == st_full["handed_back"] == 0 and st["kernel_fallback"] == st_full["kernel_fallback"]          # ... and no read leaves the staged kernels because of the band


def test_wildcard_bases_stay_on_the_staged_kernels(medium_case, env, monkeypatch):
    """A base outside A / C / G / T scores sc_N against anything (ksw2's matrix).  A problem whose read holds one, or whose target touches a
    text block that does, goes to queues of its own (bin_tasks_kernel / global_band_kernel) and is scored by the WILDC instance of dp_lane_kernel or - a queue
    with few chunks - by dp_wave_kernel, one wavefront per pair of problems.  Every read with an N, one read in five, none: the SAM text equals the oracle's
    under either kernel, no read is handed to align_kernel because of the wildcard, and dp_wave_kernel alone (every global problem forced through the full
    matrix, every extension through the tile) gives the same text as well."""
    base = list(medium_case.synth.make_reads(medium_case.pg, 3000, 150, seed=171, sub_rate=0.02, indel_rate=0.004)) + \
           list(medium_case.synth.make_reads(medium_case.pg, 1500, 250, seed=172, sub_rate=0.03, indel_rate=0.002)) + \
           list(medium_case.synth.make_reads(medium_case.pg, 1500, 90, seed=173))
    rng = np.random.default_rng(23)
    for rate in (1.0, 0.2, 0.0):
        reads = [r.copy() for r in base]
        for r in reads:
            if rng.random() < rate:
                r[int(rng.integers(0, len(r)))] = ord("N")
                if rng.random() < 0.3: r[int(rng.integers(0, len(r)))] = ord("R")
        stats = []
        for wave_max, dbg in (("96", "0"), ("0", "0"), ("1000000", "0"), ("1000000", str(65536 + 131072))):
            monkeypatch.setenv("MONI_AF_WAVE_MAX", wave_max)
            monkeypatch.setenv("MONI_AF_DBG", dbg)
            _, st = both(env, reads)
            stats.append(st)
            assert st["handed_back"] == 0 and st["handover_why"].get("wildcard_or_dirs", 0) == 0
        assert len({st["aligned"] for st in stats}) == 1 and len({st["dp_cells"] for st in stats}) == 1
        assert len({st["kernel_fallback"] for st in stats[:3]}) == 1


def test_extensions_with_more_query_than_target(medium_case, env, monkeypatch):
    """The reference cuts an extension's target at ext_len = 100 rows whatever the read's part is (aligner_ksw2.hpp:2796-2812): a read whose only seeds lie in its last (or
    first) 30-45 bases asks for an extension of 105-120 query bases against 100 target rows.  Such a problem gets its band from the diagonal over the target rows followed
    by one insertion (af_tile_band2) - few substitutions: dp_band_kernel's 16 diagonals; many, or an indel: the tile.  Both strands, so the long extension lies on either
    side; the SAM text equals the oracle's, with the bands and with every problem forced through the full tiles."""
    rng = np.random.default_rng(29)
    exact = list(medium_case.synth.make_reads(medium_case.pg, 3000, 150, seed=181, sub_rate=0.0, indel_rate=0.0))
    reads = []
    for r in exact:
        r = r.copy()
        keep = int(rng.integers(30, 46))                      # bases left exact at one end
        lo, hi = (0, 150 - keep) if rng.random() < 0.5 else (keep, 150)
        rate = (0.01, 0.04, 0.10)[int(rng.integers(0, 3))]
        for i in range(lo, hi):
            if rng.random() < rate:
                r[i] = ord("ACGT"[(("ACGT".index(chr(r[i])) if chr(r[i]) in "ACGT" else 0) + int(rng.integers(1, 4))) % 4])
        if rng.random() < 0.2 and hi - lo > 40:               # a short deletion inside the mutated part
            at = int(rng.integers(lo + 10, hi - 10)); d = int(rng.integers(1, 4))
            r = np.concatenate([r[:at], r[at + d:]])
        reads.append(r)
    _, st = both(env, reads)
    monkeypatch.setenv("MONI_AF_DBG", str(65536 + 131072))
    _, st_full = both(env, reads)
    assert st["aligned"] == st_full["aligned"] > 2000 and st["dp_cells"] == st_full["dp_cells"] and st["dp_cells_cut"] < st_full["dp_cells_cut"]
    assert st["handed_back"] == st_full["handed_back"] == 0


def test_sub_batches_and_handed_back_reads(medium_case, env, monkeypatch):
    """The batch goes through the GPU in sub-batches overlapped with the host stage; reads the kernel hands back go
    through the host pipeline and are spliced in at their positions.  Output must not depend on either."""
    reads = list(medium_case.synth.make_reads(medium_case.pg, 7001, 150, seed=151))
    monkeypatch.setenv("MONI_ALIGN_SUB", "1500")
    _, st = both(env, reads)
    assert st["handed_back"] == 0
    monkeypatch.setenv("MONI_AK_FORCE_HANDBACK", "7")
    _, st = both(env, reads)
    assert st["handed_back"] == 1001
    monkeypatch.setenv("MONI_ALIGN_SUB", "1000000")
    both(env, reads[:1])
    both(env, reads[:0])


def test_host_formatting_mode(medium_case, env, monkeypatch):
    """By default align_kernel spells the SAM lines (ak_emit) and the host only orders them; MONI_ALIGN_HOST_FORMAT=1 makes the host
    stage format them from the records (emit_record).  Both must give the oracle's text, FASTA reads (no qualities) included."""
    reads = list(medium_case.synth.make_reads(medium_case.pg, 3000, 150, seed=153, sub_rate=0.03, indel_rate=0.004))
    both(env, reads)
    both(env, reads[:500], quals=False)
    monkeypatch.setenv("MONI_ALIGN_HOST_FORMAT", "1")
    both(env, reads)
    both(env, reads[:500], quals=False)


def test_align_run_on_resident_batch(medium_case, env, monkeypatch):
    """moni_align_run (reads already in HBM) gives the text moni_align_batch gives, leaves the batch resident, also
    after reads went through the hand-back path."""
    from oracle import orc
    o, ctx = env
    reads = medium_case.synth.make_reads(medium_case.pg, 3000, 150, seed=152)
    offs = np.arange(0, 3001 * 150, 150, dtype=np.uint64)
    names, noff = orc.make_names(3000)
    q = np.full(reads.size, ord("I"), dtype=np.uint8)
    want, _ = ctx.align_batch(reads.reshape(-1), offs, names, noff, q, host_threads=4)
    ctx.upload(reads.reshape(-1), offs)
    a, st = ctx.align_run(names, noff, q, host_threads=4)
    assert a == want and st["reads"] == 3000
    monkeypatch.setenv("MONI_AK_FORCE_HANDBACK", "5")
    b, st = ctx.align_run(names, noff, q, host_threads=4)
    assert b == want and st["handed_back"] == 600
    monkeypatch.delenv("MONI_AK_FORCE_HANDBACK")
    n, st = ctx.align_run(names, noff, q, host_threads=4, want_text=False)
    assert n == len(want) and st["handed_back"] == 0


def test_sam_identical_250bp_noisy_ragged(medium_case, env):
    rng = np.random.default_rng(17)
    base = medium_case.synth.make_reads(medium_case.pg, 4000, 250, seed=9, sub_rate=0.04, indel_rate=0.006)
    reads = [r[: int(rng.integers(30, 251))].copy() for r in base]
    reads.append(np.frombuffer(b"N" * 60, dtype=np.uint8))
    reads.append(np.frombuffer(b"ACGT" * 10, dtype=np.uint8))
    x = base[0].copy(); x[40:45] = ord("N"); reads.append(x)
    reads.append(np.frombuffer(bytes(medium_case.fi.text[100:400]), dtype=np.uint8))
    y = base[1].copy(); y[:] = np.frombuffer(bytes(y).lower(), dtype=np.uint8); reads.append(y)
    both(env, reads)


def test_degenerate_reads(medium_case, env):
    """Empty and one-base reads, reads shorter than min_len, homopolymers, a read across the separator between two sequences, the very
    first and last bases of the text, IUPAC codes, exact duplicates: whatever the reference's loops make of them, both sides must agree.
    Also through moni_align_stream (lines ordered on the GPU)."""
    from oracle import orc
    text = np.frombuffer(bytes(medium_case.fi.text), dtype=np.uint8)
    ss = medium_case.fi.seq_starts
    base = list(medium_case.synth.make_reads(medium_case.pg, 200, 150, seed=77))
    edge = [np.zeros(0, np.uint8), np.frombuffer(b"A", np.uint8), np.frombuffer(b"ACGTACGTACGTACGTACGTACGT", np.uint8),       # 0, 1, 24 bases
            np.frombuffer(b"A" * 150, np.uint8), np.frombuffer(b"T" * 26, np.uint8), np.frombuffer(b"ACGTRYKMSWBDHVN" * 8, np.uint8),
            text[int(ss[1]) - 80:int(ss[1]) + 70].copy(),                        # across a separator
            text[:150].copy(), text[len(text) - 150:].copy(), text[len(text) - 40:].copy(),
            medium_case.synth.revcomp(text[None, 300:450])[0].copy(), base[3].copy(), base[3].copy()]
    rng = np.random.default_rng(5)
    reads = list(base)
    for e in edge:
        reads.insert(int(rng.integers(0, len(reads))), e)
    both(env, reads)
    both(env, reads, quals=False)
    o, ctx = env
    offs = np.zeros(len(reads) + 1, dtype=np.uint64); offs[1:] = np.cumsum([len(r) for r in reads])
    seq = np.concatenate(reads)
    names, noff = orc.make_names(len(reads))
    q = np.full(len(seq), ord("#"), dtype=np.uint8)
    want, _ = orc.align_batch(o, seq, offs, names, noff, q, threads=8)
    got, _ = ctx.align_batch(seq, offs, names, noff, q, host_threads=8, stream=True)
    assert got == want


def test_long_reads_go_through_the_hand_back_path(medium_case, env):
    """Reads whose flanks exceed the align kernel's DP capacities (query > 512) are handed back to the host pipeline by the
    kernel itself; the output must still be the oracle's."""
    rng = np.random.default_rng(23)
    base = list(medium_case.synth.make_reads(medium_case.pg, 300, 150, seed=31))
    text = medium_case.fi.text
    for ln in (600, 900, 1500):
        for _ in range(3):
            p0 = int(rng.integers(1000, len(medium_case.pg.seqs[0]) - 3000))
            r = np.array(text[p0:p0 + ln], dtype=np.uint8).copy()
            k = rng.integers(0, ln, size=ln // 60)
            r[k] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=len(k))]
            base.insert(int(rng.integers(0, len(base))), r)
    p0 = 5000
    base.insert(7, np.array(text[p0:p0 + 4200], dtype=np.uint8).copy())      # >= 4096 bases: the kernel declines it outright
    r = np.array(text[p0 + 9000:p0 + 12000], dtype=np.uint8).copy()
    r[1500] = ord("A") if r[1500] != ord("A") else ord("C")
    base.insert(50, r)
    _, st = both(env, base)
    assert st["handed_back"] >= 2


def test_one_very_long_read_does_not_fail_the_batch(medium_case, env):
    """a batch of 150 bp reads with one 7 kb read (host pipeline over the DP kernels: the oracle's record) and one 20 kb read (beyond the
    largest DP the kernels take: reported unaligned with a notice, align_reads_dispatcher.hpp:300-407 never fails a batch on one read):
    every other line is the oracle's, through both entry points"""
    from oracle import orc
    o, ctx = env
    rng = np.random.default_rng(3)
    reads = list(medium_case.synth.make_reads(medium_case.pg, 6000, 150, seed=77))
    text = medium_case.fi.text
    for at, p0, ln in ((4100, 2000, 20000), (123, 31000, 7000)):
        r = np.array(text[p0:p0 + ln], dtype=np.uint8).copy()
        k = rng.integers(0, ln, size=ln // 100)
        r[k] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=len(k))]
        reads.insert(at, r)
    offs = np.zeros(len(reads) + 1, dtype=np.uint64); offs[1:] = np.cumsum([len(r) for r in reads])
    seq = np.concatenate(reads)
    names, noff = orc.make_names(len(reads))
    q = np.full(len(seq), ord("I"), dtype=np.uint8)
    want, _ = orc.align_batch(o, seq, offs, names, noff, q, threads=8)
    wl = want.split(b"\n")
    for stream in (False, True):
        got, st = ctx.align_batch(seq, offs, names, noff, q, host_threads=8, stream=stream)
        gl = got.split(b"\n")
        assert len(gl) == len(wl)
        for i, (a, b) in enumerate(zip(gl, wl)):
            if i == 4101:                     # (the 20 kb read: inserted at 4100, then shifted by the 7 kb read at 123)
                f = a.split(b"\t")
                assert f[1] == b"4" and f[2] == b"*" and len(f[9]) == 20000, a[:80]
            else:
                assert a == b, (i, a[:200], b[:200])


def test_repeats_overflow_the_kernel_capacities(tmp_path):
    """A tandem-repeat region gives reads hundreds of seed occurrences: more anchors / chains than the align kernel's per-read
    capacities, so those reads take the host pipeline; the SAM text must not change."""
    from moni_align_amd import capi, index_build, synth
    from oracle import orc
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    unit = acgt[rng.integers(0, 4, size=180)]
    copies = []
    for _ in range(70):
        u = unit.copy()
        k = rng.integers(0, 180, size=2)
        u[k] = acgt[rng.integers(0, 4, size=2)]
        copies.append(u)
    base = np.concatenate([acgt[rng.integers(0, 4, size=15000)]] + copies + [acgt[rng.integers(0, 4, size=15000)]])
    hap = base.copy()
    k = rng.integers(0, len(hap), size=40)
    hap[k] = acgt[rng.integers(0, 4, size=40)]
    pg = synth.Pangenome(seqs=[base, hap], names=["chrR", "S1_H1_chrR"], w=10)
    fi = index_build.build_from_pangenome(pg, device="cpu")
    path = str(tmp_path / "rep.mfi")
    fi.save(path)
    reads = []
    for _ in range(150):
        p0 = int(rng.integers(14000, 15000 + 70 * 180))
        r = base[p0:p0 + 150].copy()
        k = rng.integers(0, 150, size=2)
        r[k] = acgt[rng.integers(0, 4, size=2)]
        reads.append(r if rng.random() < 0.5 else synth.revcomp(r[None, :])[0])
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        _, st = both((orc.OracleIndex(path), ctx), reads)
        assert st["handed_back"] >= 1
    finally:
        ctx.close()
        idx.close()


def test_sam_identical_on_fasta_built_index(medium_case, tmp_path):
    """The same pangenome text as a FASTA-built index (null lifts, liftidx.hpp:150-157): one RNAME per haplotype, and the CIGAR of
    column 6 still goes through levioSAM's walk (zero-length operations disappear)."""
    from moni_align_amd import capi, index_build
    from oracle import orc
    fi = index_build.build_from_pangenome(medium_case.pg, device="cpu", lifted=False)
    path = str(tmp_path / "fasta.mfi")
    fi.save(path)
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        reads = medium_case.synth.make_reads(medium_case.pg, 6000, 150, seed=154, sub_rate=0.02, indel_rate=0.003)
        both((orc.OracleIndex(path), ctx), list(reads))
        assert ctx.sam_header().count(b"@SQ") == 7
    finally:
        ctx.close()
        idx.close()


def test_lifted_records_name_the_reference_contig(medium_case, env):
    """medium_case is built `-r ref -v vcf` style: haplotypes lift onto chr19 (aligner_ksw2.hpp:3133-3160); OA keeps the haplotype."""
    from oracle import orc
    o, ctx = env
    reads = medium_case.synth.make_reads(medium_case.pg, 4000, 150, seed=155)
    offs = np.arange(0, 4001 * 150, 150, dtype=np.uint64)
    names, noff = orc.make_names(4000)
    sam, st = ctx.align_batch(reads.reshape(-1), offs, names, noff, None, host_threads=4)
    rn = [l.split(b"\t") for l in sam.split(b"\n") if l]
    assert all(f[2] in (b"chr19", b"*") for f in rn)
    oa = [[x for x in f if x.startswith(b"OA:Z:")][0][5:].split(b",")[0] for f in rn if f[2] == b"chr19"]
    assert len(set(oa)) == 7            # all seven sequences occur as the pangenome-side name
    lifted_differs = sum(1 for f in rn if f[2] == b"chr19" and f[5] != [x for x in f if x.startswith(b"OA:Z:")][0].split(b",")[3])
    assert lifted_differs > 20          # reads across haplotype indels get I / D in the lifted CIGAR


def test_sam_identical_fasta_reads(medium_case, env):
    reads = medium_case.synth.make_reads(medium_case.pg, 500, 100, seed=3)
    both(env, list(reads), quals=False)


def test_header(medium_case, env):
    o, ctx = env
    h = ctx.sam_header().decode().split("\n")
    assert h[0] == "@HD\tVN:1.6\tSO:unknown" and h[-2] == "@PG\tID:moni\tPN:moni\tVN:0.1.0"
    assert h[1] == "@SQ\tSN:chr19\tLN:%d" % len(medium_case.pg.seqs[0])


def test_csv_mem_statistics_equal_the_oracle(medium_case, env):
    """-c (moni_align_csv_batch): per read the number of MEMs, their occurrences, the extreme frequencies, the largest / smallest count of one MEM
    on one genome (count_dict, filtered occurrences included), what the filters dropped and how many chains check_left_MEM skipped - equal to
    the oracle's calculate_MEM_stats / seed_freq_filter / selection loop (aligner_ksw2.hpp:340-343, 417, 1868-1933); the SAM text is unchanged"""
    from oracle import orc
    o, ctx = env
    reads = medium_case.synth.make_reads(medium_case.pg, 3000, 150, seed=99, sub_rate=0.02)
    rl = list(reads) + [np.frombuffer(b"ACGT" * 10, np.uint8), np.frombuffer(b"N" * 70, np.uint8)]
    offs = np.zeros(len(rl) + 1, dtype=np.uint64); offs[1:] = np.cumsum([len(r) for r in rl])
    seq = np.concatenate(rl)
    names, noff = orc.make_names(len(rl))
    q = np.full(len(seq), ord("I"), np.uint8)
    sam, csv, st = ctx.align_csv_batch(seq, offs, names, noff, q, host_threads=8)
    want_sam, _ = orc.align_batch(o, seq, offs, names, noff, q, threads=8)
    assert sam == want_sam
    want_csv = orc.align_csv(o, seq, offs, names, noff)
    if csv != want_csv:
        g, w = csv.split(b"\n"), want_csv.split(b"\n")
        i = next(k for k in range(min(len(g), len(w))) if g[k] != w[k])
        raise AssertionError("CSV differs at line %d: got %r want %r" % (i, g[i], w[i]))
    cols = np.array([[float(x) for x in ln.split(b",")[1:]] for ln in csv.split(b"\n") if ln])
    assert (cols[:, 7] > 0).any() and (cols[:, 4] >= cols[:, 5]).all() and (cols[:, 0] > 0).sum() > 2900      # chains were skipped somewhere; high >= low
