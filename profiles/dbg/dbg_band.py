"""debugging aid: records of the first reads of tests/test_gpu_align.py::test_sam_identical_150bp under the banding switches against the oracle"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moni_align_amd import capi, synth, index_build
from oracle import orc

pg = synth.make_pangenome(60000, 6, site_spacing=700)
fi = index_build.build_from_pangenome(pg, device="cpu")
d = tempfile.mkdtemp(); path = os.path.join(d, "m.mfi"); fi.save(path)
if os.environ.get("DBG_MIX"):
    rl = list(synth.make_reads(pg, 6000, 150, seed=161, sub_rate=0.02, indel_rate=0.004)) + list(synth.make_reads(pg, 3000, 250, seed=162, sub_rate=0.03, indel_rate=0.002)) + list(synth.make_reads(pg, 3000, 100, seed=163))
    N = len(rl)
    offs = np.zeros(N + 1, dtype=np.uint64); offs[1:] = np.cumsum([len(r) for r in rl])
    seq = np.concatenate(rl)
else:
    reads = synth.make_reads(pg, 20000, 150, seed=150)
    N = int(os.environ.get("DBG_N", "2000"))
    reads = reads[:N]
    offs = np.arange(0, (N + 1) * 150, 150, dtype=np.uint64)
    seq = reads.reshape(-1)
nrate = float(os.environ.get("DBG_N_RATE", "0"))
if nrate > 0:      # a wildcard base in that share of the reads
    seq = seq.copy(); rng = np.random.default_rng(7)
    for r in np.nonzero(rng.random(N) < nrate)[0]:
        seq[int(offs[r]) + int(rng.integers(0, int(offs[r + 1] - offs[r])))] = ord("N")
names, noff = orc.make_names(N)
q = np.full(len(seq), ord("I"), dtype=np.uint8)
want, _ = orc.align_batch(orc.OracleIndex(path), seq, offs, names, noff, q, threads=8)
idx = capi.Index(fi=fi); ctx = capi.Ctx(idx)
wl = want.split(b"\n")
SW = ("MONI_AF_DBG", "MONI_AF_NOPLANK", "MONI_AF_WAVE_MAX", "MONI_AF_FIN_V1")
for cfg in ({}, {"MONI_AF_FIN_V1": "1"}, {"MONI_AF_NOPLANK": "1"}, {"MONI_AF_DBG": "131072"}, {"MONI_AF_DBG": "65536"}, {"MONI_AF_DBG": "196608"}, {"MONI_AF_WAVE_MAX": "0"}, {"MONI_AF_WAVE_MAX": "1000000"},
            {"MONI_AF_WAVE_MAX": "1000000", "MONI_AF_DBG": "196608"}):
    for k in SW: os.environ.pop(k, None)
    os.environ.update(cfg)
    got, st = ctx.align_batch(seq, offs, names, noff, q, host_threads=8)
    gl = got.split(b"\n")
    bad = [k for k, (a, b) in enumerate(zip(gl, wl)) if a != b]
    print("%s: %d differing records of %d, %d to align_kernel, %d to host; cells %d cut %d slots %d; why %s" % (cfg, len(bad), N, st["kernel_fallback"], st["handed_back"], st["dp_cells"], st["dp_cells_cut"], st["dp_slots"], st["handover_why"]), bad[:20])
    for k in bad[:3]:
        ga, wa = gl[k].split(b"\t"), wl[k].split(b"\t")
        print("  got :", b" ".join(ga[:9] + ga[11:]).decode()[:400]); print("  want:", b" ".join(wa[:9] + wa[11:]).decode()[:400])
