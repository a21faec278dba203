"""moni_align_run (in-order gather on the GPU) vs moni_align_batch vs the oracle on a small lifted index, with several sub-batches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moni_align_amd import capi, index_build, synth
from oracle import orc

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L = 150
pg = synth.make_pangenome(60000, 6, site_spacing=700)
fi = index_build.build_from_pangenome(pg, device="cpu")
idx = capi.Index(fi=fi); ctx = capi.Ctx(idx)
reads = synth.make_reads(pg, n_reads, L, seed=150)
offs = np.arange(0, (n_reads + 1) * L, L, dtype=np.uint64)
names, noff = orc.make_names(n_reads)
q = np.full(reads.size, ord("I"), dtype=np.uint8)
want, cnt = orc.align_batch(orc.OracleIndex(fi=fi), reads.reshape(-1), offs, names, noff, q, threads=8)
for sub in ("", "3000", "1000000"):
    if sub: os.environ["MONI_ALIGN_SUB"] = sub
    ctx.upload(reads.reshape(-1), offs)
    sam, st = ctx.align_run(names, noff, q, host_threads=8)
    print("sub", sub or "default", "identical:", sam == want, {k: st[k] for k in ("reads", "aligned", "handed_back", "kernel_fallback")}, flush=True)
    sam2, st = ctx.align_run(names, noff, q, host_threads=8)
    print("  again:", sam2 == want)
os.environ["MONI_ALIGN_SUB"] = "3000"; os.environ["MONI_SEED_EST"] = "0.5,2"          # pipelined seeding with buffers that must grow mid-batch
sam, st = ctx.align_run(names, noff, q, host_threads=8)
print("growth path identical:", sam == want, flush=True)
del os.environ["MONI_SEED_EST"]
os.environ["MONI_SEED_WHOLE"] = "1"
sam, st = ctx.align_run(names, noff, q, host_threads=8)
print("whole-batch seeding identical:", sam == want, flush=True)
del os.environ["MONI_SEED_WHOLE"]
sam, st = ctx.align_batch(reads.reshape(-1), offs, names, noff, q, host_threads=8)
print("align_batch (host order, pipelined) identical:", sam == want, flush=True)
sam, st = ctx.align_batch(reads.reshape(-1), offs, names, noff, q, host_threads=8, stream=True)
print("align_stream identical:", sam == want, len(sam), len(want), flush=True)
if sam != want:
    la, lb = sam.split(b"\n"), want.split(b"\n")
    for k, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            print("first difference at record", k, "\n got:", x[:300], "\nwant:", y[:300], flush=True)
            break
ctx.upload(reads.reshape(-1), offs)
os.environ["MONI_ALIGN_HOST_ORDER"] = "1"
sam, st = ctx.align_run(names, noff, q, host_threads=8)
print("host order identical:", sam == want)
