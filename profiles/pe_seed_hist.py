"""What the paired plan kernel's LDS instance has to hold: per pair, the seeds of both mates and their occurrences (anchors), before and after the
direction filter's better half (an upper bound of what pe_plan_kernel keeps).  Index from bench.py's cache (MONI_BENCH_SAVE_INDEX=1)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moni_align_amd import capi, synth

pg = synth.make_pangenome(61420004, 12, seed=19, var_seed=12)
idx = capi.Index(path="/tmp/moni_bench_cache/idx_61420004_12_lifted_0.mfi", device=0)
N, L = 100000, 150
mates, _ = synth.make_pairs(pg, N, L, seed=350)
ctx = capi.Ctx(idx)
ctx.upload(mates.reshape(-1), np.arange(0, (2 * N + 1) * L, L, dtype=np.uint64))
ctx.seed_run(min_len=25, filter_seeds=True, n_seeds_thr=5000)
s = ctx.seed_fetch()
mems, rmo = s["mems"], s["read_mem_off"].astype(np.int64)
print("fields", mems.dtype.names)
occ = mems["occ_cnt"].astype(np.int64)
rc = (mems["mate"] & 2) != 0
read_of = np.repeat(np.arange(2 * N), np.diff(rmo))
pair_of = read_of // 2
is_m2 = (read_of & 1) == 1
dir1 = (~is_m2 & ~rc) | (is_m2 & rc)          # mate 1 forward + mate 2 reverse-complemented
n_raw = np.bincount(pair_of, minlength=N)
a_all = np.bincount(pair_of, weights=occ, minlength=N)
a_d1 = np.bincount(pair_of, weights=occ * dir1, minlength=N)
a_best = np.maximum(a_d1, a_all - a_d1)
n_d1 = np.bincount(pair_of, weights=dir1, minlength=N)
n_best = np.where(a_d1 >= a_all - a_d1, n_d1, n_raw - n_d1)
for name, v in (("seeds, both directions", n_raw), ("anchors, both directions", a_all), ("seeds, larger direction", n_best), ("anchors, larger direction", a_best)):
    print("%-28s mean %7.1f  p50 %5d  p90 %5d  p99 %5d  p99.9 %6d  max %6d" % (name, v.mean(), *[int(np.percentile(v, q)) for q in (50, 90, 99, 99.9)], int(v.max())))
for cap_raw, cap_an in ((64, 96), (80, 128), (96, 128), (96, 160), (128, 192), (160, 256)):
    print("raw <= %3d and anchors(all) <= %3d: %.4f of the pairs;  anchors(larger direction) <= %3d: %.4f" % (cap_raw, cap_an, np.mean((n_raw <= cap_raw) & (a_all <= cap_an)), cap_an, np.mean((n_raw <= cap_raw) & (a_best <= cap_an))))
ctx.close(); idx.close()
