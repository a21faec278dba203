"""The C-ABI library loads and exports every symbol include/moni_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

from moni_align_amd import capi


def test_exports_match_header():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "moni_hip.h")).read()
    declared = set(re.findall(r"\b(moni_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    capi.build_lib()
    L = ctypes.CDLL(capi.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in capi.lib().moni_version()


def test_no_device_fails_loudly(small_case):
    import torch
    if torch.cuda.is_available():
        return
    try:
        capi.Index(fi=small_case.fi)
    except RuntimeError as e:
        assert "moni_index_create" in str(e)
    else:
        raise AssertionError("index creation must fail without a HIP device")
