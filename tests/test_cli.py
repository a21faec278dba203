"""moni-hip-align: the reference's command line (src/align/align_full_ksw2.cpp:101-431).  Without a GPU only argument
and FASTA/FASTQ parsing can be exercised (--dry-run); the end-to-end run is a -m gpu test."""
import gzip
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "moni_align_amd", "host", "moni-hip-align")


@pytest.fixture(scope="module")
def exe():
    import __graft_entry__
    __graft_entry__.build()
    return EXE


def test_dry_run_parsing(exe, small_case, tmp_path):
    reads = small_case.synth.make_reads(small_case.pg, 37, 100, seed=2)
    fq = str(tmp_path / "r.fastq")
    small_case.synth.write_fastq(fq, reads)
    out = subprocess.check_output([exe, "idx/pref", "-p", fq, "-t", "3", "-l", "30", "-S", "1000", "-F", "0.5", "-O", "4,13", "-E", "2,1",
                                   "--gpus", "2", "--dry-run"]).decode()
    assert "reads=37 bases=3700" in out and "min_len=30" in out and "S=1000 F=0.50 O=4,13 E=2,1 threads=3 gpus=2" in out
    assert ("out=%s_pref_30.sam" % fq) in out and "first=simulated.0" in out
    # gzip + multi-line FASTA
    fa = str(tmp_path / "r.fa.gz")
    with gzip.open(fa, "wb") as f:
        for i, r in enumerate(reads[:5]):
            b = r.tobytes()
            f.write(b">q%d some comment\n%s\n%s\n" % (i, b[:60], b[60:]))
    out = subprocess.check_output([exe, "x", "-p", fa, "-o", "o.sam", "--dry-run"]).decode()
    assert "reads=5 bases=500" in out and "out=o.sam" in out and "first=q0" in out
    # struct defaults of the reference binary when the wrapper does not pass -S/-F
    assert "S=5000 F=0.30" in out


def test_dry_run_write_multi_gpu_batching(exe, small_case, tmp_path):
    """--gpus 2 --dry-run-write: no GPU, but the single-end front end's multi-GPU batching runs - the mapped file cut into record-aligned ranges, the workers of
    both (absent) GPUs taking ranges, every block written at its place in the file - with placeholder records: the output holds every read once, in input
    order, and more than one worker of each GPU wrote into it (align_reads_dispatcher.hpp:201-294 is the shape this replaces)."""
    reads = small_case.synth.make_reads(small_case.pg, 6000, 100, seed=5)
    fq = str(tmp_path / "r.fastq")
    small_case.synth.write_fastq(fq, reads)
    out = str(tmp_path / "o.sam")
    r = subprocess.run([exe, "idx/pref", "-p", fq, "-o", out, "--gpus", "2", "--gpu-batch", "250", "-t", "4", "--dry-run-write"], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert b"reads=6000 bases=600000" in r.stdout and b"gpus=2" in r.stdout
    lines = open(out, "rb").read().split(b"\n")
    assert lines[-1] == b"" and len(lines) == 6001
    workers = set()
    for i, ln in enumerate(lines[:-1]):
        f = ln.split(b"\t")
        assert f[0] == b"simulated.%d" % i and f[1] == b"4" and f[9] == reads[i].tobytes() and len(f[10]) == 100
        workers.add(int(f[11][5:]))
    assert len(workers) >= 3 and min(workers) < 3 <= max(workers)          # three contexts per GPU: workers 0-2 are GPU 0's, 3-5 GPU 1's


def test_unsupported_modes_exit_1(exe, tmp_path):
    fq = str(tmp_path / "x.fq")
    open(fq, "w").write("@a\nACGT\n+\nIIII\n")
    for extra, msg in ((["-1", fq], b"needs both -1 and -2"), (["-1", fq, "-2", fq, "--ms"], b"take single-end input"), ([], b"no reads given")):
        r = subprocess.run([exe, "x"] + extra, capture_output=True)
        assert r.returncode == 1 and msg in r.stderr
    assert subprocess.run([exe], capture_output=True).returncode == 1


@pytest.mark.gpu
def test_cli_end_to_end(exe, medium_case, tmp_path):
    from oracle import orc
    N, L = 3000, 150
    reads = medium_case.synth.make_reads(medium_case.pg, N, L, seed=44)
    fq = str(tmp_path / "reads.fastq")
    medium_case.synth.write_fastq(fq, reads)
    prefix = medium_case.path[:-4]
    out = str(tmp_path / "out.sam")
    subprocess.check_call([exe, prefix, "-p", fq, "-o", out, "-S", "1000", "-F", "0.5", "-t", "4", "--gpu-batch", "1000"])
    o = orc.OracleIndex(medium_case.path)
    offs = np.arange(0, (N + 1) * L, L, dtype=np.uint64)
    names, noff = orc.make_names(N)
    want, _ = orc.align_batch(o, reads.reshape(-1), offs, names, noff, np.full(N * L, ord("I"), dtype=np.uint8), with_header=True, threads=8)
    assert open(out, "rb").read() == want


@pytest.mark.gpu
def test_cli_report_mems_legacy_outputs_and_mouse_fixture_reads(exe, medium_case, tmp_path):
    """-m (report-MEMs SAM), --ms / --mems (legacy `moni ms` / `moni mems` text files) against the oracle, with several batches in
    flight; and the reference's own read file data/mouse/reads (names with /1, real quality strings) as a parser / format fixture."""
    from oracle import orc
    N, L = 2500, 150
    reads = medium_case.synth.make_reads(medium_case.pg, N, L, seed=45)
    fq = str(tmp_path / "reads.fastq")
    medium_case.synth.write_fastq(fq, reads)
    prefix = medium_case.path[:-4]
    o = orc.OracleIndex(medium_case.path)
    offs = np.arange(0, (N + 1) * L, L, dtype=np.uint64)
    names, noff = orc.make_names(N)
    quals = np.full(N * L, ord("I"), dtype=np.uint8)
    out = str(tmp_path / "mems.sam")
    subprocess.check_call([exe, prefix, "-p", fq, "-o", out, "-m", "-S", "1000", "-F", "0.5", "--gpu-batch", "700"])
    hdr = orc.align_batch(o, reads[:0].reshape(-1), offs[:1], names[:0], noff[:1], None, with_header=True)[0]
    assert open(out, "rb").read() == hdr + orc.report_mems_batch(o, reads.reshape(-1), offs, names, noff, quals)
    base = str(tmp_path / "legacy")
    subprocess.check_call([exe, prefix, "-p", fq, "-o", base, "--ms", "--gpu-batch", "900"])
    subprocess.check_call([exe, prefix, "-p", fq, "-o", base, "--mems", "--gpu-batch", "900"])
    wp, wl, wm = [], [], []
    for i in range(N):
        p, l = o.ms_lengths(reads[i].tobytes())
        h = ">simulated.%d\n" % i
        wp.append(h + "".join("%d " % x for x in p) + "\n")
        wl.append(h + "".join("%d " % x for x in l) + "\n")
        wm.append(h + "".join("(%d,%d) " % t for t in orc.legacy_mems(p, l)) + "\n")
    assert open(base + ".pointers").read() == "".join(wp)
    assert open(base + ".lengths").read() == "".join(wl)
    assert open(base + ".mems").read() == "".join(wm)


def test_reference_fixture_reads_parse(exe):
    """data/mouse/reads/mouse.chr19.R1.fastq of the reference (5000 x 100 bp, names with /1): both readers give the same batch."""
    src = "/root/reference/data/mouse/reads/mouse.chr19.R1.fastq"
    if not os.path.exists(src):
        pytest.skip("the reference checkout is not on this machine")
    out = subprocess.check_output([exe, "x", "-p", src, "--dry-run"]).decode()
    assert "reads=5000 bases=500000" in out
    first = open(src).readline().split()[0][1:]
    assert ("first=" + first) in out


def read_fastq(path):
    names, seqs, quals = [], [], []
    with open(path, "rb") as f:
        while True:
            h = f.readline()
            if not h:
                break
            names.append(h[1:].split()[0]); seqs.append(f.readline().rstrip(b"\n")); f.readline(); quals.append(f.readline().rstrip(b"\n"))
    return names, seqs, quals


@pytest.mark.gpu
def test_cli_on_the_reference_mouse_reads_planted_in_a_pangenome(exe, tmp_path):
    """The reference's own read file (data/mouse/reads/mouse.chr19.R1.fastq, copied as data to tests/golden/ref_data/mouse: 5000 x 100 bp
    with real quality strings and /1 names).  No mouse reference sequence exists here, so the first 800 reads are planted (every other one
    reverse-complemented, every fifth with two substitutions) 300 bp apart in a random base genome from which 4 haplotypes are derived:
    the CLI's SAM file for all 5000 reads (planted ones align, the others mostly do not) equals the oracle's."""
    from moni_align_amd import index_build, synth
    from oracle import orc
    src = os.path.join(ROOT, "tests", "golden", "ref_data", "mouse", "mouse.chr19.R1.fastq")
    names, seqs, quals = read_fastq(src)
    assert len(names) == 5000 and all(len(s) == 100 for s in seqs)
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    base = acgt[rng.integers(0, 4, size=800 * 300 + 500)].copy()
    for k in range(800):
        r = np.frombuffer(seqs[k], np.uint8).copy()
        r[~np.isin(r, acgt)] = ord("A")               # an N of a read cannot be planted in an ACGT index
        if k % 2:
            r = synth.revcomp(r[None, :])[0]
        if k % 5 == 0:
            r[[20, 70]] = acgt[(np.searchsorted(acgt, r[[20, 70]]) + 1) & 3]
        base[200 + 300 * k:300 + 300 * k] = r
    pg = synth.make_pangenome(len(base), 4, site_spacing=900, base=base)
    fi = index_build.build_from_pangenome(pg, device="cpu")
    mfi = str(tmp_path / "mouse.mfi")
    fi.save(mfi)
    out = str(tmp_path / "mouse.sam")
    subprocess.check_call([exe, mfi[:-4], "-p", src, "-o", out, "-S", "1000", "-F", "0.5", "-t", "4", "--gpu-batch", "1800"])
    seq = np.frombuffer(b"".join(seqs), np.uint8)
    offs = np.arange(0, 5001 * 100, 100, dtype=np.uint64)
    nm = np.frombuffer(b"".join(names), np.uint8)
    noff = np.zeros(5001, np.uint64); noff[1:] = np.cumsum([len(x) for x in names])
    want, cnt = orc.align_batch(orc.OracleIndex(fi=fi), seq, offs, nm, noff, np.frombuffer(b"".join(quals), np.uint8), with_header=True, threads=8)
    got = open(out, "rb").read()
    assert got == want
    assert cnt["aligned"] >= 790


@pytest.mark.gpu
def test_cli_paired_end(exe, medium_case, tmp_path):
    """-1 / -2 -u: st_align's paired loop (learn on batches of -b pairs, align them, then the rest) through the binary, against the
    oracle's paired path (oracle/align_pe.hpp); one mate file gzip-compressed (the kseq-style reader), the other plain (mmap)."""
    from oracle import orc
    from tests.test_host_sim_pe import oracle_pe
    from tests.test_oracle_pe import make_pairs
    m1, m2, _ = make_pairs(medium_case.pg, 1500, L=100)
    f1, f2 = str(tmp_path / "m_1.fastq"), str(tmp_path / "m_2.fastq.gz")
    with open(f1, "wb") as f:
        for i, r in enumerate(m1):
            f.write(b"@p%d/1 first mate\n%s\n+\n%s\n" % (i, r.tobytes(), b"I" * len(r)))
    with gzip.open(f2, "wb") as f:
        for i, r in enumerate(m2):
            f.write(b"@p%d/2\n%s\n+\n%s\n" % (i, r.tobytes(), b"I" * len(r)))
    prefix = medium_case.path[:-4]
    out = str(tmp_path / "pe.sam")
    log = subprocess.check_output([exe, prefix, "-1", f1, "-2", f2, "-u", "-o", out, "-S", "1000", "-F", "0.5", "-t", "4", "-b", "512", "--gpu-batch", "2048"]).decode()
    o = orc.OracleIndex(medium_case.path)
    want, st = oracle_pe(o, m1, m2, b_size=512)
    hdr, _ = orc.align_batch(o, np.zeros(0, np.uint8), np.zeros(1, np.uint64), np.zeros(0, np.uint8), np.zeros(1, np.uint64), None, with_header=True)
    got = open(out, "rb").read()
    assert got[:len(hdr)] == hdr
    if got[len(hdr):] != want:
        from tests.test_host_sim_pe import first_diff
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got[len(hdr):], want))
    assert ("Number of aligned pairs: %d/1500" % st["aligned"]) in log
    # without -u: orphan recovery on (the reference's default)
    out2 = str(tmp_path / "pe_orphan.sam")
    subprocess.check_call([exe, prefix, "-1", f1, "-2", f2, "-o", out2, "-S", "1000", "-F", "0.5", "-t", "4", "-b", "512", "--gpu-batch", "2048"])
    want2, _ = oracle_pe(o, m1, m2, b_size=512, find_orphan=True)
    assert open(out2, "rb").read()[len(hdr):] == want2
    # -m: the MEM records of the pairs (several batches through the reader thread / two workers / in-place writes)
    out3 = str(tmp_path / "pe_mems.sam")
    f2p = str(tmp_path / "m_2.fastq")             # both files plain: the mapped reader (newlines counted and records parsed by several threads, mates interleaved)
    with open(f2p, "wb") as f:
        f.write(gzip.open(f2, "rb").read())
    subprocess.check_call([exe, prefix, "-1", f1, "-2", f2p, "-m", "-o", out3, "-S", "1000", "-F", "0.5", "-t", "4", "--gpu-batch", "600"])
    want3, _ = oracle_pe(o, m1, m2, b_size=512, report_mems=True)
    got3 = open(out3, "rb").read()[len(hdr):]
    if got3 != want3:
        from tests.test_host_sim_pe import first_diff
        raise AssertionError("-m records differ at record %d:\n got: %s\nwant: %s" % first_diff(got3, want3))
    # -Z: secondary chains
    out5 = str(tmp_path / "pe_z.sam")
    subprocess.check_call([exe, prefix, "-1", f1, "-2", f2p, "-Z", "-o", out5, "-S", "1000", "-F", "0.5", "-t", "4", "-b", "512", "--gpu-batch", "2048"])
    want5, _ = oracle_pe(o, m1, m2, b_size=512, find_orphan=True, secondary_chains=True)
    assert want5 != want2 and open(out5, "rb").read()[len(hdr):] == want5
    # -c: one line of MEM statistics per pair in <sam>.csv, the SAM file unchanged (several batches, several workers: the lines in input order)
    out6 = str(tmp_path / "pe_csv.sam")
    subprocess.check_call([exe, prefix, "-1", f1, "-2", f2p, "-u", "-c", "-o", out6, "-S", "1000", "-F", "0.5", "-t", "4", "-b", "512", "--gpu-batch", "700"])
    want6, _ = oracle_pe(o, m1, m2, b_size=512, csv=True)
    assert open(out6, "rb").read()[len(hdr):] == want
    got6 = open(out6 + ".csv", "rb").read()
    assert got6 == b"Read,Unique,Total,Max_Freq,Min_Freq,Highest_Occ,Lowest_Occ,Filtered,Chains_Skipped\n" + want6 and got6.count(b"\n") == 1501
    out4 = str(tmp_path / "pe_plain.sam")
    subprocess.check_call([exe, prefix, "-1", f1, "-2", f2p, "-o", out4, "-S", "1000", "-F", "0.5", "-t", "4", "-b", "512", "--gpu-batch", "300"])
    assert open(out4, "rb").read()[len(hdr):] == want2


def test_dry_run_paired(exe, tmp_path):
    """-1 / -2 argument handling and the two readers (plain + gzip) without a GPU"""
    f1, f2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq.gz")
    with open(f1, "w") as f:
        f.write("".join("@x%d/1 c\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(7)))
    with gzip.open(f2, "wt") as f:
        f.write("".join("@x%d/2\nTTGCA\n+\nIIIII\n" % i for i in range(7)))
    out = subprocess.check_output([exe, "idx/pref", "-1", f1, "-2", f2, "-u", "-d", "-D", "40", "-b", "256", "-l", "30", "--dry-run"]).decode()
    assert "pairs=7 bases=105" in out and "filter_dir=0 dir_thr=40.0 find_orphan=0 b=256" in out and "first=x0/1 second=x0/2" in out
    assert ("out=%s_pref_30.sam" % f1) in out
    out = subprocess.check_output([exe, "idx/pref", "-1", f1, "-2", f2, "--dry-run"]).decode()
    assert "filter_dir=1 dir_thr=50.0 find_orphan=1 b=512" in out
    # different numbers of records
    with open(f1, "a") as f:
        f.write("@extra/1\nACGT\n+\nIIII\n")
    r = subprocess.run([exe, "idx/pref", "-1", f1, "-2", f2, "--dry-run"], capture_output=True)
    assert r.returncode == 1 and b"different numbers of records" in r.stderr
    # only one mate file
    r = subprocess.run([exe, "idx/pref", "-1", f1, "--dry-run"], capture_output=True)
    assert r.returncode == 1 and b"needs both -1 and -2" in r.stderr


@pytest.mark.gpu
def test_cli_csv_mem_statistics(exe, medium_case, tmp_path):
    """-c: <sam>.csv with one line of MEM statistics per read (calculate_MEM_stats, seed_freq_filter's and check_left_MEM's counts:
    aligner_ksw2.hpp:340-343, 417, 1868-1933; csv.hpp:55-67) beside the unchanged SAM file - both equal the oracle's; -n -q are accepted
    when the index files of those forms are there"""
    from oracle import orc
    N, L = 1500, 150
    reads = medium_case.synth.make_reads(medium_case.pg, N, L, seed=46, sub_rate=0.02)
    fq = str(tmp_path / "reads.fastq")
    medium_case.synth.write_fastq(fq, reads)
    prefix = medium_case.path[:-4]
    out = str(tmp_path / "out.sam")
    subprocess.check_call([exe, prefix, "-p", fq, "-o", out, "-S", "3", "-F", "0.5", "-t", "4", "-c", "--gpu-batch", "600"])
    o = orc.OracleIndex(medium_case.path)
    offs = np.arange(0, (N + 1) * L, L, dtype=np.uint64)
    names, noff = orc.make_names(N)
    # (the CLI passes -S 3 to the seeding: the oracle's capi takes the same threshold through its config defaults only for 1000, so compare through the library call)
    from moni_align_amd import capi
    idx = capi.Index(path=medium_case.path); ctx = capi.Ctx(idx)
    sam, csv, st = ctx.align_csv_batch(reads.reshape(-1), offs, names, noff, np.full(N * L, ord("I"), np.uint8), host_threads=4, n_seeds_thr=3)
    ctx.close(); idx.close()
    hdr = b"Read,Unique,Total,Max_Freq,Min_Freq,Highest_Occ,Lowest_Occ,Filtered,Chains_Skipped\n"
    assert open(out + ".csv", "rb").read() == hdr + csv
    body = open(out, "rb").read()
    assert body.endswith(sam) and body[:3] == b"@HD"
