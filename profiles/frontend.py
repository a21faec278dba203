"""moni-hip-align end to end on the GPU box: FASTQ file -> SAM file, at the bench's index (BASELINE.json configs[2] shape).
Usage (through gpurun, from the repo root): python3 profiles/frontend.py [n_reads] [extra CLI args...]
Writes the FASTQ (synthetic reads, names simulated.<i>, quality I) to /tmp, runs the CLI, and checks the first 50,000 records of the
SAM file against moni_align_batch on the same reads (which tests/test_gpu_fullsize.py proves equal to the oracle)."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moni_align_amd import capi, index_build, synth

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
extra = sys.argv[2:]
L = 150
cache = "/tmp/moni_bench_cache"
mfi = os.path.join(cache, "idx_61420004_12_lifted_0.mfi")
t0 = time.time()
pg = synth.make_pangenome(61420004, 12, seed=19, var_seed=12)
if not os.path.exists(mfi):
    import torch
    os.makedirs(cache, exist_ok=True)
    fi = index_build.build_from_pangenome(pg, device="cuda:0")
    fi.save(mfi)
    del fi
    torch.cuda.empty_cache()
print("index ready after %.0fs" % (time.time() - t0), flush=True)
reads = synth.make_reads(pg, n_reads, L, seed=150)
fq = "/tmp/frontend_reads.fq"
t1 = time.time()
with open(fq, "wb") as f:          # records in blocks of equal name length
    lo = 0
    while lo < n_reads:
        digits = len(str(lo))
        hi = min(n_reads, 10 ** digits)
        k = hi - lo
        name = np.frombuffer(("@simulated." + "0" * digits + "\n").encode(), np.uint8)
        rec = np.empty((k, len(name) + L + 1 + 2 + L + 1), np.uint8)
        rec[:, :len(name)] = name
        idx = np.arange(lo, hi)
        for d in range(digits):
            rec[:, len(name) - 2 - d] = (idx // 10 ** d) % 10 + 48
        o = len(name)
        rec[:, o:o + L] = reads[lo:hi]; rec[:, o + L] = 10
        rec[:, o + L + 1] = ord("+"); rec[:, o + L + 2] = 10
        rec[:, o + L + 3:o + 2 * L + 3] = ord("I"); rec[:, o + 2 * L + 3] = 10
        f.write(rec.tobytes())
        lo = hi
print("FASTQ: %d reads, %.2f GB, written in %.0fs" % (n_reads, os.path.getsize(fq) / 1e9, time.time() - t1), flush=True)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "moni_align_amd", "host", "moni-hip-align")
out = "/tmp/frontend_out.sam"
cmd = [exe, mfi[:-4], "-p", fq, "-o", out, "-t", "16", "-S", "1000", "-F", "0.5"] + extra
print(" ".join(cmd), flush=True)
t2 = time.time()
r = subprocess.run(cmd, capture_output=True, text=True)
wall = time.time() - t2
print(r.stdout[-3000:], r.stderr[-6000:], flush=True)
print("CLI wall %.2fs (index load included) -> %.2f M reads/s end to end; SAM %.2f GB" % (wall, n_reads / wall / 1e6, os.path.getsize(out) / 1e9), flush=True)
# identity of the head of the file with the library call the parity tests cover
m = 50000
idx = capi.Index(path=mfi); ctx = capi.Ctx(idx)
offs = np.arange(0, (m + 1) * L, L, dtype=np.uint64)
names = ("".join("simulated.%d" % i for i in range(m))).encode()
noff = np.zeros(m + 1, np.uint64); noff[1:] = np.cumsum([len("simulated.%d" % i) for i in range(m)])
q = np.full(m * L, ord("I"), np.uint8)
want, _ = ctx.align_batch(reads[:m].reshape(-1), offs, np.frombuffer(names, np.uint8), noff, q, host_threads=8)
with open(out, "rb") as f:
    body = b"".join(l for l in f.read(len(want) + (1 << 20)).split(b"\n") if False) if False else None
    f.seek(0)
    data = f.read(len(want) + (1 << 16))
hdr_end = 0
while data[hdr_end:hdr_end + 1] == b"@":
    hdr_end = data.index(b"\n", hdr_end) + 1
print("first %d records identical to moni_align_batch: %s" % (m, data[hdr_end:hdr_end + len(want)] == want), flush=True)
