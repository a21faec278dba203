"""Independent properties of the SAM text (tests/sam_props.py) on the oracle's output (CPU) and on the HIP path's output at the
BASELINE sizes (-m gpu)."""
import numpy as np
import pytest

from tests import sam_props


def test_oracle_sam_satisfies_independent_properties(medium_case):
    from oracle import orc
    o = orc.OracleIndex(medium_case.path)
    reads = medium_case.synth.make_reads(medium_case.pg, 3000, 150, seed=161, sub_rate=0.02, indel_rate=0.004)
    offs = np.arange(0, 3001 * 150, 150, dtype=np.uint64)
    names, noff = orc.make_names(3000)
    sam, cnt = orc.align_batch(o, reads.reshape(-1), offs, names, noff, None, threads=4)
    st = sam_props.check_records(sam, medium_case.pg.seqs, medium_case.pg.names, reads)
    assert st["aligned"] == cnt["aligned"] and st["as_checked"] > 2500


@pytest.mark.gpu
@pytest.mark.timeout(1500)
def test_hip_sam_properties_at_mouse_scale():
    from moni_align_amd import capi, synth
    from oracle import orc
    from tests.test_gpu_fullsize import build_or_load
    pg, fi = build_or_load(61420004, 12)
    n = 200000
    reads = synth.make_reads(pg, n, 150, seed=977)
    offs = np.arange(0, (n + 1) * 150, 150, dtype=np.uint64)
    names, noff = orc.make_names(n)
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    try:
        sam, st = ctx.align_batch(reads.reshape(-1), offs, names, noff, None, host_threads=16)
    finally:
        ctx.close()
        idx.close()
    r = sam_props.check_records(sam, pg.seqs, pg.names, reads, stride=20)
    assert r["aligned"] > 9900 and r["as_checked"] > 9000
