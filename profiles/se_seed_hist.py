"""Seeds and anchors per read (what chain_plan_kernel's LDS instances have to hold) for a bench configuration; index from bench.py's cache.
python profiles/se_seed_hist.py BASE_LEN HAPS READ_LEN"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moni_align_amd import capi, synth

base_len, haps, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pg = synth.make_pangenome(base_len, haps, seed=19, var_seed=12)
idx = capi.Index(path="/tmp/moni_bench_cache/idx_%d_%d_lifted_0.mfi" % (base_len, haps), device=0)
N = 100000
reads = synth.make_reads(pg, N, L, seed=150)
ctx = capi.Ctx(idx)
ctx.upload(reads.reshape(-1), np.arange(0, (N + 1) * L, L, dtype=np.uint64))
ctx.seed_run(min_len=25, filter_seeds=True, n_seeds_thr=1000)
s = ctx.seed_fetch()
mems, rmo = s["mems"], s["read_mem_off"].astype(np.int64)
occ = mems["occ_cnt"].astype(np.int64)
read_of = np.repeat(np.arange(N), np.diff(rmo))
n_raw = np.bincount(read_of, minlength=N)
tot = np.bincount(read_of, weights=occ, minlength=N)
keep = occ / np.maximum(1, tot[read_of]) <= 0.5          # seed_freq_filter, freq_thr 0.5
n_mem = np.bincount(read_of, weights=keep, minlength=N)
anch = np.bincount(read_of, weights=occ * keep, minlength=N)
for name, v in (("seeds (MEMs + halves)", n_raw), ("seeds after the frequency filter", n_mem), ("anchors", anch)):
    print("%-34s mean %7.1f  p50 %5d  p90 %5d  p99 %5d  p99.9 %6d  max %6d" % (name, v.mean(), *[int(np.percentile(v, q)) for q in (50, 90, 99, 99.9)], int(v.max())))
for cm, ca in ((24, 96), (32, 160), (48, 192), (64, 256), (64, 320), (96, 384), (96, 512), (256, 2048)):
    print("seeds <= %3d and anchors <= %4d: %.4f of the reads" % (cm, ca, np.mean((n_mem <= cm) & (anch <= ca))))
ctx.close(); idx.close()
