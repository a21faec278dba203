"""Writes the small paired-end fixture: 160 FR pairs (clean, noisy, one mate without a 25-base MEM, one mate of noise) on a 20 kbp x 3-haplotype
synthetic pangenome built `-r ref -v vcf` style, and what the CPU oracle's paired path (oracle/align_pe.hpp) writes for them without and
with orphan recovery, plus the learnt fragment model.  Guards the oracle (its compile flags included: the model's doubles are compared
exactly), the host replay of pe_core.h and, under -m gpu, the HIP path against silent drift; NOT an output of the reference binary."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from moni_align_amd import index_build, synth   # noqa: E402
from oracle import orc                           # noqa: E402


def inputs():
    pg = synth.make_pangenome(20000, 3, seed=23, var_seed=5, site_spacing=500)
    rng = np.random.default_rng(61)
    m1, m2 = [], []
    for i in range(160):
        s = pg.seqs[int(rng.integers(0, len(pg.seqs)))]
        ins = int(max(230, rng.normal(330, 25)))
        p = int(rng.integers(0, len(s) - ins))
        frag = s[p:p + ins].copy()
        k = rng.random(ins) < 0.01
        frag[k] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(k.sum()))]
        a, b = frag[:100].copy(), synth.revcomp(frag[None, ins - 100:])[0].copy()
        if rng.random() < 0.5:
            a, b = b, a
        if i % 8 == 3:                           # no 25-base MEM in mate 2 (or mate 1): a case for orphan recovery
            x = (b if (i // 8) % 2 == 0 else a)
            # a substitution every 19 bases: the recovered mate scores far above its minimum 20 + 8 ln(100) = 56; every 5: just above (80);
            # every 4: below (50) - recovery finds the place but the mate stays unmapped (aligner_ksw2.hpp:2471,2519)
            for q in range(9 if (i // 8) % 3 == 0 else 2, 100, (19, 4, 5)[(i // 8) % 3]):
                x[q] = ord("A") if x[q] != ord("A") else ord("C")
        if i % 16 == 5:
            b = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=100)].copy()
        if i % 20 == 7:
            a = a[:70].copy(); b = b[:85].copy()
        m1.append(a); m2.append(b)
    return pg, m1, m2


def oracle_runs(fi, m1, m2):
    n = len(m1)
    o1 = np.zeros(n + 1, np.uint64); o1[1:] = np.cumsum([len(x) for x in m1])
    o2 = np.zeros(n + 1, np.uint64); o2[1:] = np.cumsum([len(x) for x in m2])
    nm1 = [b"g%d/1" % i for i in range(n)]; nm2 = [b"g%d/2" % i for i in range(n)]
    no1 = np.zeros(n + 1, np.uint64); no1[1:] = np.cumsum([len(x) for x in nm1])
    no2 = np.zeros(n + 1, np.uint64); no2[1:] = np.cumsum([len(x) for x in nm2])
    q1 = np.full(int(o1[-1]), ord("I"), np.uint8); q2 = np.full(int(o2[-1]), ord("I"), np.uint8)
    o = orc.OracleIndex(fi=fi)
    out = {}
    for orphan in (False, True):
        out[orphan] = orc.align_pe(o, np.concatenate(m1), o1, np.concatenate(m2), o2, np.frombuffer(b"".join(nm1), np.uint8), no1,
                                   np.frombuffer(b"".join(nm2), np.uint8), no2, q1, q2, b_size=512, find_orphan=orphan)
    return out


def main():
    pg, m1, m2 = inputs()
    fi = index_build.build_from_pangenome(pg, device="cpu", lifted=True)
    runs = oracle_runs(fi, m1, m2)
    for orphan, name in ((False, "pe_small.sam"), (True, "pe_small_orphan.sam")):
        with open(os.path.join(HERE, name), "wb") as f:
            f.write(runs[orphan][0])
    st = runs[True][1]
    with open(os.path.join(HERE, "pe_small_model.json"), "w") as f:
        json.dump({"ins_count": st["ins_count"], "ins_mean_hex": float(st["ins_mean"]).hex(), "ins_std_dev_hex": float(st["ins_std_dev"]).hex(),
                   "aligned_without_orphan": runs[False][1]["aligned"], "aligned_with_orphan": st["aligned"],
                   "orphan_pairs": st["orphan_pairs"], "orphan_recovered": st["orphan_recovered"]}, f, indent=1)
    print("wrote", runs[False][0].count(b"\n"), "SAM lines;", st)


if __name__ == "__main__":
    main()
