import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        import torch
        return torch.cuda.device_count() > 0          # counting devices does not initialise the GPU
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests are skipped (not failed) on a box without a HIP device; on a GPU box they run and fail loudly when the
    HIP library is missing (the product has no CPU path)."""
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device on this box")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


class SmallCase:
    """A small synthetic pangenome + its flat index for brute-force checks."""

    def __init__(self, tmpdir, base_len, n_haps, site_spacing, tag):
        from moni_align_amd import synth, index_build
        self.pg = synth.make_pangenome(base_len, n_haps, site_spacing=site_spacing)
        self.fi = index_build.build_from_pangenome(self.pg, device="cpu")
        self.path = os.path.join(str(tmpdir), tag + ".mfi")
        self.fi.save(self.path)
        self.text = self.fi.text.tobytes()
        self.synth = synth


@pytest.fixture(scope="session")
def small_case(tmp_path_factory):
    return SmallCase(tmp_path_factory.mktemp("idx"), 4000, 4, 250, "small")


@pytest.fixture(scope="session")
def medium_case(tmp_path_factory):
    return SmallCase(tmp_path_factory.mktemp("idx"), 60000, 6, 700, "medium")
