"""The product's host pipeline (align_host.hpp: filter, chaining, chain-selection rounds, CIGAR stitching, MAPQ, SAM)
driven by CPU stand-ins for the kernels, against the oracle's sequential restatement of aligner::align: SAM text must be
byte-identical.  The same comparison runs with the real kernels under -m gpu (tests/test_gpu_align.py)."""
import numpy as np
import pytest

from oracle import orc
from tests.host_sim import sim as hs


def first_diff(a: bytes, b: bytes):
    la, lb = a.split(b"\n"), b.split(b"\n")
    for k, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return k, x.decode()[:400], y.decode()[:400]
    return min(len(la), len(lb)), "<len %d>" % len(la), "<len %d>" % len(lb)


def run_case(case, reads_list, threads=3):
    offs = np.zeros(len(reads_list) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads_list])
    seq = np.concatenate(reads_list)
    names, noff = orc.make_names(len(reads_list))
    quals = np.full(len(seq), ord("I"), dtype=np.uint8)
    o = orc.OracleIndex(case.path)
    want, wcnt = orc.align_batch(o, seq, offs, names, noff, quals, threads=2)
    got, st = hs.Sim(case.fi).align_batch(seq, offs, names, noff, quals, threads=threads)
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))
    return want, wcnt, st


def test_sam_identical_150bp(medium_case):
    reads = medium_case.synth.make_reads(medium_case.pg, 1500, 150, seed=150)
    want, wcnt, st = run_case(medium_case, list(reads))
    assert wcnt["aligned"] > 1400 and int(st[1]) == wcnt["aligned"]
    assert int(st[3]) >= wcnt["dp_cells"] // 4


def test_sam_identical_noisy_and_ragged(medium_case):
    rng = np.random.default_rng(17)
    base = medium_case.synth.make_reads(medium_case.pg, 600, 250, seed=9, sub_rate=0.04, indel_rate=0.006)
    reads = [r[: int(rng.integers(30, 251))].copy() for r in base]
    reads.append(np.frombuffer(b"N" * 60, dtype=np.uint8))
    reads.append(np.frombuffer(b"ACGT" * 10, dtype=np.uint8))
    x = base[0].copy(); x[40:45] = ord("N"); reads.append(x)
    reads.append(np.frombuffer(bytes(medium_case.fi.text[100:400]), dtype=np.uint8))
    y = base[1].copy(); y[:] = np.frombuffer(bytes(y).lower(), dtype=np.uint8); reads.append(y)
    run_case(medium_case, reads)


def test_sam_identical_without_quals(small_case):
    reads = small_case.synth.make_reads(small_case.pg, 100, 120, seed=4)
    offs = np.arange(0, 101 * 120, 120, dtype=np.uint64)
    names, noff = orc.make_names(100, "r")
    o = orc.OracleIndex(small_case.path)
    want, _ = orc.align_batch(o, reads.reshape(-1), offs, names, noff, None)
    got, _ = hs.Sim(small_case.fi).align_batch(reads.reshape(-1), offs, names, noff, None, threads=2)
    assert got == want


def test_device_side_logic_replayed_on_host(medium_case):
    """align_core.h (what lane 0 of the align kernel runs: STL-free chaining with the libstdc++ sort emulation, selection
    loop, fill_chain, CIGAR stitching) must give the oracle's SAM too; reads that overflow its capacities are reported."""
    rng = np.random.default_rng(5)
    reads = list(medium_case.synth.make_reads(medium_case.pg, 1200, 150, seed=77))
    base = medium_case.synth.make_reads(medium_case.pg, 300, 250, seed=9, sub_rate=0.04, indel_rate=0.006)
    reads += [r[: int(rng.integers(30, 251))].copy() for r in base]
    reads.append(np.frombuffer(b"N" * 60, dtype=np.uint8))
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    seq = np.concatenate(reads)
    names, noff = orc.make_names(len(reads))
    quals = np.full(len(seq), ord("I"), dtype=np.uint8)
    o = orc.OracleIndex(medium_case.path)
    want, wcnt = orc.align_batch(o, seq, offs, names, noff, quals, threads=2)
    got, st = hs.Sim(medium_case.fi).align_core_batch(seq, offs, names, noff, quals)
    assert int(st[3]) == 0, "reads overflowed the device capacities"
    if got != want:
        raise AssertionError("SAM differs at record %d:\n got: %s\nwant: %s" % first_diff(got, want))
    assert int(st[1]) == wcnt["aligned"]
