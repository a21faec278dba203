"""What ms_lf_kernel reads beyond one fast row per LF step (VERDICT r2, item 3 ii): the -DMONI_MS_ATTR builds of the library count, in the spare bits of the jump
counter, 1: the next-run walks (one more fast row each), 2: the entries into the general path.  MONI_HIP_LIB selects the build; index from bench.py's cache."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moni_align_amd import capi, synth

pg = synth.make_pangenome(61420004, 12, seed=19, var_seed=12)
idx = capi.Index(path="/tmp/moni_bench_cache/idx_61420004_12_lifted_0.mfi", device=0)
N, L = 1000000, 150
reads = synth.make_reads(pg, N, L, seed=150)
ctx = capi.Ctx(idx)
ctx.upload(reads.reshape(-1), np.arange(0, (N + 1) * L, L, dtype=np.uint64))
ctx.seed_run(min_len=25, filter_seeds=True, n_seeds_thr=1000)
c = ctx.counters()
S, Jraw = int(c[0]), int(c[1])
print("%s: S = %d LF steps, J = %d threshold jumps, extra = %d (%.3f per step)" % (os.environ.get("MONI_HIP_LIB", "product build"), S, Jraw & ((1 << 28) - 1), Jraw >> 28, (Jraw >> 28) / S))
ctx.close(); idx.close()
