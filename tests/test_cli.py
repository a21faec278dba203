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


def test_unsupported_modes_exit_1(exe, tmp_path):
    fq = str(tmp_path / "x.fq")
    open(fq, "w").write("@a\nACGT\n+\nIIII\n")
    for extra in (["-1", fq, "-2", fq], ["-p", fq, "-m"], ["-p", fq, "-q"]):
        r = subprocess.run([exe, "x"] + extra, capture_output=True)
        assert r.returncode == 1 and b"not implemented" in r.stderr
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
