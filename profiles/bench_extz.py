#!/usr/bin/env python3
"""Stand-alone throughput of the two forms of the extension DP (register-blocked extz_kernel<NCH>, LDS-tiled
extz_lds_kernel = what align_kernel runs) on uniform batches of score-only problems: GCUPS from the library's HIP events.
Usage (GPU box): python3 profiles/bench_extz.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moni_align_amd import capi, index_build, synth          # noqa: E402


def main():
    pg = synth.make_pangenome(200000, 2, seed=5, var_seed=6)
    fi = index_build.build_from_pangenome(pg, device="cuda:0", log=lambda *a: None)
    idx = capi.Index(fi=fi)
    rng = np.random.default_rng(1)
    for form in ("registers", "lds"):
        os.environ["MONI_EXTZ_LDS"] = "1" if form == "lds" else "0"
        ctx = capi.Ctx(idx)
        for (ql, tl, flag) in ((62, 100, 1), (125, 100, 1), (20, 100, 1), (8, 8, 1), (150, 250, 1), (62, 100, 0x42)):
            n = max(2000, min(400000, int(2.5e9 / (ql * tl))))
            q = rng.integers(0, 4, size=n * ql, dtype=np.uint8)
            t = rng.integers(0, 4, size=n * tl, dtype=np.uint8)
            tasks = np.zeros(n, dtype=capi.DP_TASK_DTYPE)
            tasks["q_off"] = np.arange(n, dtype=np.uint64) * ql
            tasks["t_off"] = np.arange(n, dtype=np.uint64) * tl
            tasks["qlen"] = ql; tasks["tlen"] = tl; tasks["flag"] = flag
            ctx.extz_batch(q, t, tasks)
            ctx.extz_batch(q, t, tasks)
            ms = ctx.kernel_ms(5)
            print("%-9s q=%3d t=%3d flag=0x%02x  n=%6d  kernel %.3f ms  %.1f GCUPS  %.2f us/problem/CU-wave-slot" %
                  (form, ql, tl, flag, n, ms, n * ql * tl / ms / 1e6, ms * 1e3 / n), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
