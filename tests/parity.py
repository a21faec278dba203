"""Shared comparison of a seeding result (HIP path or its host replay) with the oracle's."""
import numpy as np


def assert_seeds_equal(got, want):
    """got: {'mems': MEM_DTYPE array, 'occs', 'read_mem_off'}; want: oracle.orc seed_batch dict."""
    m = got["mems"]
    assert len(m) == len(want["pos"]), (len(m), len(want["pos"]))
    assert np.array_equal(got["read_mem_off"], want["read_mem_off"])
    for f in ("pos", "len", "idx", "mate", "rpos", "total_occ", "num_filtered", "occ_cnt", "read", "occ_off"):
        a = m[f].astype(np.uint64)
        b = want[f]
        if not np.array_equal(a, b):
            k = int(np.nonzero(a != b)[0][0])
            raise AssertionError("field %s differs first at MEM %d: got %d want %d" % (f, k, a[k], b[k]))
    assert np.array_equal(got["occs"], want["occs"])
