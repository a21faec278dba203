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
