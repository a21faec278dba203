"""moni-hip-align -1 / -2 -u on synthetic FASTQ files (FR pairs, 2 x 150 bp, 1 Mbp x 8 haplotypes): pairs per second of the whole run
(index load excluded: the binary's own "Elapsed time" starts after it).  python profiles/pe_frontend.py [--pairs 200000]"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=200000)
    args = ap.parse_args()
    import __graft_entry__
    __graft_entry__.build()
    from moni_align_amd import index_build, synth
    from tests.test_oracle_pe import make_pairs
    pg = synth.make_pangenome(1000000, 8, seed=19, var_seed=12)
    fi = index_build.build_from_pangenome(pg, device="cuda:0")
    m1, m2, _ = make_pairs(pg, args.pairs, L=150, seed=3)
    with tempfile.TemporaryDirectory() as d:
        fi.save(os.path.join(d, "idx.mfi"))
        for k, mm in ((1, m1), (2, m2)):
            with open(os.path.join(d, "m_%d.fastq" % k), "wb") as f:
                q = b"I" * 150
                f.write(b"".join(b"@p%d/%d\n%s\n+\n%s\n" % (i, k, r.tobytes(), q) for i, r in enumerate(mm)))
        exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "moni_align_amd", "host", "moni-hip-align")
        for rep in range(2):
            t0 = time.time()
            out = subprocess.check_output([exe, os.path.join(d, "idx"), "-1", os.path.join(d, "m_1.fastq"), "-2", os.path.join(d, "m_2.fastq"), "-u", "-o",
                                           os.path.join(d, "out.sam"), "-S", "1000", "-F", "0.5", "-t", "16"]).decode()
            print("run %d (%.2f s wall incl. index load, SAM %d MB):" % (rep, time.time() - t0, os.path.getsize(os.path.join(d, "out.sam")) >> 20))
            print("".join(l + "\n" for l in out.splitlines() if "pairs" in l.lower() or "Elapsed" in l or "Insert" in l))


if __name__ == "__main__":
    main()
