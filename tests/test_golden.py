"""Committed fixtures (tests/golden): the index builder, the oracle and (under -m gpu) the HIP path must keep
reproducing them."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def golden():
    import sys
    sys.path.insert(0, G)
    import make_golden_small
    from moni_align_amd import index_build
    pg, reads = make_golden_small.inputs()
    fi = index_build.build_from_pangenome(pg, device="cpu", lifted=False)
    z = np.load(os.path.join(G, "seed_small.npz"))
    return fi, reads, z


@pytest.fixture(scope="module")
def golden_lifted():
    import sys
    sys.path.insert(0, G)
    import make_golden_small
    from moni_align_amd import index_build
    pg, reads = make_golden_small.inputs()
    return index_build.build_from_pangenome(pg, device="cpu", lifted=True), reads


def test_lifted_index_fixture(golden_lifted):
    """The same text built `-r ref -v vcf` style (haplotypes lift onto the reference contig): oracle and the product's host
    pipeline reproduce the committed lifted SAM; every record names the reference contig."""
    from oracle import orc
    from tests.host_sim import sim as hs
    fi, reads = golden_lifted
    offs = np.arange(0, 65 * 120, 120, dtype=np.uint64)
    names, noff = orc.make_names(64)
    quals = np.full(64 * 120, ord("I"), dtype=np.uint8)
    want = open(os.path.join(G, "align_small_lifted.sam"), "rb").read()
    sam, _ = orc.align_batch(orc.OracleIndex(fi=fi), reads.reshape(-1), offs, names, noff, quals, with_header=True)
    assert sam == want
    body = b"".join(l + b"\n" for l in want.split(b"\n") if l and not l.startswith(b"@"))
    got, _ = hs.Sim(fi).align_batch(reads.reshape(-1), offs, names, noff, quals, threads=2)
    assert got == body
    assert all(l.split(b"\t")[2] in (b"chr19", b"*") for l in body.split(b"\n") if l)


def test_builder_reproduces_fixture(golden):
    fi, reads, z = golden
    assert np.array_equal(reads, z["reads"])
    for k in ("text", "heads", "starts", "ssa", "esa", "thr", "slcp"):
        assert np.array_equal(getattr(fi, k), z[k]), k


def test_oracle_reproduces_fixture(golden):
    from oracle import orc
    fi, reads, z = golden
    o = orc.OracleIndex(fi=fi)
    offs = np.arange(0, 65 * 120, 120, dtype=np.uint64)
    s = o.seed_batch(reads.reshape(-1), offs, 25, True, 1000)
    for k, v in s.items():
        assert np.array_equal(v, z["seed_" + k]), k
    names, noff = orc.make_names(64)
    quals = np.full(64 * 120, ord("I"), dtype=np.uint8)
    sam, _ = orc.align_batch(o, reads.reshape(-1), offs, names, noff, quals, with_header=True)
    assert sam == open(os.path.join(G, "align_small.sam"), "rb").read()


def test_host_pipeline_reproduces_sam_fixture(golden):
    from oracle import orc
    from tests.host_sim import sim as hs
    fi, reads, z = golden
    offs = np.arange(0, 65 * 120, 120, dtype=np.uint64)
    names, noff = orc.make_names(64)
    quals = np.full(64 * 120, ord("I"), dtype=np.uint8)
    sam, _ = hs.Sim(fi).align_batch(reads.reshape(-1), offs, names, noff, quals, threads=2)
    want = open(os.path.join(G, "align_small.sam"), "rb").read()
    body = b"".join(l + b"\n" for l in want.split(b"\n") if l and not l.startswith(b"@"))
    assert sam == body


@pytest.mark.gpu
def test_hip_path_reproduces_fixtures(golden):
    from moni_align_amd import capi
    from oracle import orc
    from tests.parity import assert_seeds_equal
    fi, reads, z = golden
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    offs = np.arange(0, 65 * 120, 120, dtype=np.uint64)
    ctx.upload(reads.reshape(-1), offs)
    ctx.seed_run(25, True, 1000)
    got = ctx.seed_fetch()
    want = {k[5:]: z[k] for k in z.files if k.startswith("seed_")}
    assert_seeds_equal(got, want)
    names, noff = orc.make_names(64)
    quals = np.full(64 * 120, ord("I"), dtype=np.uint8)
    sam, _ = ctx.align_batch(reads.reshape(-1), offs, names, noff, quals)
    ref = open(os.path.join(G, "align_small.sam"), "rb").read()
    assert ctx.sam_header() + sam == ref
    ctx.close(); idx.close()


@pytest.mark.gpu
def test_hip_path_reproduces_lifted_fixture(golden_lifted):
    from moni_align_amd import capi
    from oracle import orc
    fi, reads = golden_lifted
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    offs = np.arange(0, 65 * 120, 120, dtype=np.uint64)
    names, noff = orc.make_names(64)
    quals = np.full(64 * 120, ord("I"), dtype=np.uint8)
    sam, _ = ctx.align_batch(reads.reshape(-1), offs, names, noff, quals)
    assert ctx.sam_header() + sam == open(os.path.join(G, "align_small_lifted.sam"), "rb").read()
    ctx.close(); idx.close()


# ---- paired-end fixture (tests/golden/make_golden_pe.py) ----
@pytest.fixture(scope="module")
def golden_pe():
    import json
    import sys
    sys.path.insert(0, G)
    import make_golden_pe
    from moni_align_amd import index_build
    pg, m1, m2 = make_golden_pe.inputs()
    fi = index_build.build_from_pangenome(pg, device="cpu", lifted=True)
    want = {False: open(os.path.join(G, "pe_small.sam"), "rb").read(), True: open(os.path.join(G, "pe_small_orphan.sam"), "rb").read()}
    return make_golden_pe, fi, m1, m2, want, json.load(open(os.path.join(G, "pe_small_model.json")))


def _pe_interleaved(m1, m2):
    reads = [x for p in zip(m1, m2) for x in p]
    offs = np.zeros(len(reads) + 1, np.uint64); offs[1:] = np.cumsum([len(r) for r in reads])
    nm = [b"g%d/%d" % (i, k + 1) for i in range(len(m1)) for k in range(2)]
    noff = np.zeros(len(nm) + 1, np.uint64); noff[1:] = np.cumsum([len(x) for x in nm])
    seq = np.concatenate(reads)
    return seq, offs, np.frombuffer(b"".join(nm), np.uint8), noff, np.full(len(seq), ord("I"), np.uint8)


def test_paired_fixture_oracle_and_host_replay(golden_pe):
    """the oracle's paired path reproduces the committed SAM text (with and without orphan recovery) and the fragment model bit for bit
    (the model is compared as hex doubles: a change of the oracle's floating-point compile flags shows here), and the host replay of
    pe_core.h + pe_host.hpp gives the same text"""
    from tests.host_sim import sim as hs
    mk, fi, m1, m2, want, model = golden_pe
    runs = mk.oracle_runs(fi, m1, m2)
    for orphan in (False, True):
        assert runs[orphan][0] == want[orphan]
    st = runs[True][1]
    assert float(st["ins_mean"]).hex() == model["ins_mean_hex"] and float(st["ins_std_dev"]).hex() == model["ins_std_dev_hex"]
    assert st["ins_count"] == model["ins_count"] and st["orphan_recovered"] == model["orphan_recovered"] and st["aligned"] == model["aligned_with_orphan"]
    seq, offs, names, noff, q = _pe_interleaved(m1, m2)
    S = hs.Sim(fi)
    for orphan in (False, True):
        got, stats = S.align_pe_batch(seq, offs, names, noff, q, finalize=True, mean=st["ins_mean"], std_dev=st["ins_std_dev"], find_orphan=orphan)
        assert int(stats[3]) == 0 and got == want[orphan]


@pytest.mark.gpu
def test_paired_fixture_gpu(golden_pe):
    from moni_align_amd import capi
    mk, fi, m1, m2, want, model = golden_pe
    seq, offs, names, noff, q = _pe_interleaved(m1, m2)
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        m = ctx.pe_learn(seq, offs)
        assert float(m.mean).hex() == model["ins_mean_hex"] and float(m.std_dev).hex() == model["ins_std_dev_hex"] and m.count == model["ins_count"]
        for orphan in (False, True):
            got, st = ctx.pe_align(seq, offs, names, noff, q, m, host_threads=4, find_orphan=int(orphan))
            assert got == want[orphan]
            assert st["aligned"] == (model["aligned_with_orphan"] if orphan else model["aligned_without_orphan"])
    finally:
        ctx.close(); idx.close()
