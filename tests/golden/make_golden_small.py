"""Writes the small seeded fixtures: a 6 kbp x 4-haplotype synthetic pangenome, 64 reads, and what the CPU oracle
computes for them (seeds and SAM).  Guards against silent drift of the oracle, the index builder and the HIP path; it is
NOT an output of the reference binary (which cannot be built here), so parity with upstream stays unpinned."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from moni_align_amd import index_build, synth   # noqa: E402
from oracle import orc                           # noqa: E402


def inputs():
    pg = synth.make_pangenome(6000, 4, seed=7, var_seed=3, site_spacing=300)
    reads = synth.make_reads(pg, 64, 120, seed=11, sub_rate=0.02, indel_rate=0.002)
    reads[5, 30] = ord("N")
    reads[6, 10:14] = np.frombuffer(b"acgt", dtype=np.uint8)
    return pg, reads


def main():
    pg, reads = inputs()
    fi = index_build.build_from_pangenome(pg, device="cpu", lifted=False)      # FASTA-built form: null lifts
    o = orc.OracleIndex(fi=fi)
    offs = np.arange(0, 65 * 120, 120, dtype=np.uint64)
    seeds = o.seed_batch(reads.reshape(-1), offs, 25, True, 1000)
    names, noff = orc.make_names(64)
    quals = np.full(64 * 120, ord("I"), dtype=np.uint8)
    sam, cnt = orc.align_batch(o, reads.reshape(-1), offs, names, noff, quals, with_header=True)
    np.savez_compressed(os.path.join(HERE, "seed_small.npz"), reads=reads, text=fi.text, heads=fi.heads, starts=fi.starts,
                        ssa=fi.ssa, esa=fi.esa, thr=fi.thr, slcp=fi.slcp,
                        **{"seed_" + k: v for k, v in seeds.items()})
    with open(os.path.join(HERE, "align_small.sam"), "wb") as f:
        f.write(sam)
    # the same text as a `-r ref -v vcf` build: haplotypes lift onto the reference contig (liftidx.hpp:89-95,159-164)
    fl = index_build.build_from_pangenome(pg, device="cpu", lifted=True)
    sam_l, _ = orc.align_batch(orc.OracleIndex(fi=fl), reads.reshape(-1), offs, names, noff, quals, with_header=True)
    with open(os.path.join(HERE, "align_small_lifted.sam"), "wb") as f:
        f.write(sam_l)
    print("wrote", len(seeds["pos"]), "MEMs,", len(seeds["occs"]), "occurrences,", sam.count(b"\n"), "SAM lines")


if __name__ == "__main__":
    main()
