"""lsort::sort (moni_align_amd/csrc/sort_emul.h) must permute exactly like libstdc++ std::sort, ties included: the
reference's unstable sorts (chain.hpp:246,402) decide which of several equally good chains is reported."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sort_emulation_matches_std_sort(tmp_path):
    exe = str(tmp_path / "sort_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "host_sim", "sort_test.cpp")])
    out = subprocess.check_output([exe]).decode()
    assert out.startswith("OK"), out
