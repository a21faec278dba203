"""K contexts on one GPU, each with 1/K of the 1 M-read batch resident, one caller thread each: does the step get shorter when the seeding
of one context runs beside the align kernels of another?  (the index comes from bench.py's cache: run it once with MONI_BENCH_SAVE_INDEX=1)"""
import json, os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moni_align_amd import capi, index_build, synth

pg = synth.make_pangenome(61420004, 12, seed=19, var_seed=12)
idx = capi.Index(path="/tmp/moni_bench_cache/idx_61420004_12_lifted_0.mfi", device=0)
N, L = 1000000, 150
reads = synth.make_reads(pg, N, L, seed=150)
names, noff = synth.make_names(N)
quals = np.full(N * L, ord("I"), np.uint8)
out = {}
for K in [int(x) for x in (sys.argv[1:] or ["1", "2", "3", "4"])]:
    ctxs, parts = [], []
    for k in range(K):
        a, b = N * k // K, N * (k + 1) // K
        c = capi.Ctx(idx)
        c.upload(reads[a:b].reshape(-1), np.arange(0, (b - a + 1) * L, L, dtype=np.uint64))
        ctxs.append(c)
        parts.append((names[int(noff[a]):int(noff[b])], (noff[a:b + 1] - noff[a]).astype(np.uint64), quals[a * L:b * L]))
    T = max(1, 16 // K)
    def work(k, reps):
        for _ in range(reps):
            ctxs[k].align_run(*parts[k], host_threads=T, want_text=False)
    for reps in (2, 8):
        th = [threading.Thread(target=work, args=(k, reps)) for k in range(K)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = (time.perf_counter() - t0) / reps
    out[K] = {"ms_per_step": dt * 1e3, "reads_per_s": N / dt}
    print(K, out[K], flush=True)
    for c in ctxs: c.close()
print(json.dumps(out))
