"""moni-hip-align -1 / -2 (orphan recovery on) over two synthetic FASTQ files of FR pairs, 2 x 150 bp, on the configs[2] index from bench.py's cache
(MONI_BENCH_SAVE_INDEX=1): pairs per second of the whole run as the binary reports it (index load excluded: its "Elapsed time" starts after it), and
the first records against the library call.  python profiles/pe_frontend.py [--pairs 2000000]"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8000000)
    args = ap.parse_args()
    from moni_align_amd import synth
    pg = synth.make_pangenome(61420004, 12, seed=19, var_seed=12)
    prefix = "/tmp/moni_bench_cache/idx_61420004_12_lifted_0"
    assert os.path.exists(prefix + ".mfi"), "run bench.py once with MONI_BENCH_SAVE_INDEX=1"
    N, L = args.pairs, 150
    mates, _ = synth.make_pairs(pg, N, L, seed=350)
    d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        q = b"I" * L
        for k in (1, 2):
            with open(os.path.join(d, "m_%d.fastq" % k), "wb") as f:
                for lo in range(0, N, 100000):
                    f.write(b"".join(b"@simulated.%d/%d\n%s\n+\n%s\n" % (p, k, mates[2 * p + k - 1].tobytes(), q) for p in range(lo, min(N, lo + 100000))))
        exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "moni_align_amd", "host", "moni-hip-align")
        for rep in range(2):
            t0 = time.time()
            out = subprocess.check_output([exe, prefix, "-1", os.path.join(d, "m_1.fastq"), "-2", os.path.join(d, "m_2.fastq"), "-o", os.path.join(d, "out.sam"), "-S", "1000", "-F", "0.5", "-t", "16"], stderr=subprocess.STDOUT, env=dict(os.environ, MONI_CLI_VERBOSE="1")).decode()
            print("run %d (%.2f s wall incl. index load, SAM %d MB):" % (rep, time.time() - t0, os.path.getsize(os.path.join(d, "out.sam")) >> 20), flush=True)
            print("".join(l + "\n" for l in out.splitlines() if "pairs" in l.lower() or "Elapsed" in l or "Insert" in l or "Stage" in l or l.startswith("batch ")), flush=True)
        # the first 20000 pairs of the file against the library call with the same model (st_align's order: learnt on 2 batches of 512, then aligned)
        from moni_align_amd import capi
        idx = capi.Index(path=prefix + ".mfi", device=0)
        ctx = capi.Ctx(idx)
        names, noff = synth.make_pair_names(N)
        model = capi.PeModelC()
        at = 0
        while not model.complete:
            e = at + 512
            ctx.pe_learn(mates[2 * at:2 * e].reshape(-1), np.arange(0, (2 * 512 + 1) * L, L, dtype=np.uint64), model)
            at = e
        n0 = 20000
        want, _ = ctx.pe_align(mates[:2 * n0].reshape(-1), np.arange(0, (2 * n0 + 1) * L, L, dtype=np.uint64), names[:int(noff[2 * n0])], noff[:2 * n0 + 1],
                               np.full(2 * n0 * L, ord("I"), np.uint8), model, host_threads=16)
        with open(os.path.join(d, "out.sam"), "rb") as f:
            txt = f.read(64 << 20)
        body = txt[txt.index(b"\n@PG"):]
        body = body[body.index(b"\n", 1) + 1:]
        got = b"\n".join(body.split(b"\n")[:2 * n0]) + b"\n"
        print("first %d pairs of the file identical to the library call: %s" % (n0, got == want))
        ctx.close(); idx.close()
    finally:
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        os.rmdir(d)


if __name__ == "__main__":
    main()
