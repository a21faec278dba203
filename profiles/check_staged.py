"""Staged align kernels vs the oracle on a small lifted index: SAM identity and how many reads take the general kernel."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moni_align_amd import capi, index_build, synth
from oracle import orc

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
pg = synth.make_pangenome(60000, 6, site_spacing=700)
fi = index_build.build_from_pangenome(pg, device="cpu")
idx = capi.Index(fi=fi); ctx = capi.Ctx(idx)
reads = synth.make_reads(pg, n_reads, L, seed=150)
offs = np.arange(0, (n_reads + 1) * L, L, dtype=np.uint64)
names, noff = orc.make_names(n_reads)
q = np.full(reads.size, ord("I"), dtype=np.uint8)
sam, st = ctx.align_batch(reads.reshape(-1), offs, names, noff, q, host_threads=8)
print({k: st[k] for k in ("reads", "aligned", "dp_tasks", "dp_cells", "handed_back", "kernel_fallback", "dp_reused")})
want, cnt = orc.align_batch(orc.OracleIndex(fi=fi), reads.reshape(-1), offs, names, noff, q, threads=8)
print("identical:", sam == want, cnt["dp_calls"], cnt["dp_cells"])
