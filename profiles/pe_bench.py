"""Paired-end throughput of moni_pe_learn_batch / moni_pe_align_batch on one GPU (first paired path: one pair per lane in
pe_align_kernel, host finishing; find_orphan at its default, on).  Synthetic FR pairs, insert 350 +- 30, 150 bp mates, 0.5 % substitutions.
    python profiles/pe_bench.py [--pairs 100000] [--base-len 1000000] [--haps 8] [--check 2000]
Prints one JSON line; --check N compares the first N pairs with the oracle (CPU)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=100000)
    ap.add_argument("--base-len", type=int, default=1000000)
    ap.add_argument("--haps", type=int, default=8)
    ap.add_argument("--len", type=int, default=150)
    ap.add_argument("--check", type=int, default=2000)
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    from moni_align_amd import capi, index_build, synth
    from tests.test_host_sim_pe import interleave
    from tests.test_oracle_pe import make_pairs
    pg = synth.make_pangenome(args.base_len, args.haps, seed=19, var_seed=12)
    t0 = time.time()
    fi = index_build.build_from_pangenome(pg, device="cuda:0")
    t_index = time.time() - t0
    m1, m2, _ = make_pairs(pg, args.pairs, L=args.len, seed=3)
    seq, offs, names, noff, q = interleave(m1, m2)
    idx = capi.Index(fi=fi)
    ctx = capi.Ctx(idx)
    model = capi.PeModelC()
    # learn on batches of 512 pairs, as the reference does
    at = 0
    t0 = time.time()
    while not model.complete and at < args.pairs:
        hi = min(args.pairs, at + 512)
        ctx.pe_learn(seq[int(offs[2 * at]):int(offs[2 * hi])], offs[2 * at:2 * hi + 1] - offs[2 * at], model)
        at = hi
    t_learn = time.time() - t0
    ctx.pe_align(seq[:int(offs[2048])], offs[:2049], names[:int(noff[2048])], noff[:2049], q[:int(offs[2048])], model, host_threads=args.threads)      # warm-up
    t0 = time.time()
    sam, st = ctx.pe_align(seq, offs, names, noff, q, model, host_threads=args.threads)
    dt = time.time() - t0
    res = {"metric": "aligned read pairs per second (paired-end, orphan recovery on)", "value": args.pairs / dt, "unit": "pairs/s", "pairs": args.pairs,
           "seconds": dt, "t_seed": st["t_seed"], "t_kernel_and_copies": st["t_dp"], "t_host_finish": st["t_host"], "aligned": st["aligned"],
           "dp_tasks": st["dp_tasks"], "dp_cells": st["dp_cells"], "model": {"count": model.count, "mean": model.mean, "std_dev": model.std_dev},
           "t_learn": t_learn, "t_index": t_index, "workload": "%d bp x %d haplotypes, %d x 2 x %d bp" % (args.base_len, args.haps, args.pairs, args.len)}
    if args.check:
        from oracle import orc
        from tests.test_host_sim_pe import oracle_pe
        n = min(args.check, args.pairs)
        o = orc.OracleIndex(fi=fi)
        t0 = time.time()
        want, ost = oracle_pe(o, m1[:n], m2[:n], b_size=512, find_orphan=True)
        res["oracle_pairs_per_s_1thread"] = n / (time.time() - t0)
        # the oracle learnt on the same first batches when n >= the pairs the model needed
        got = b"\n".join(sam.split(b"\n")[:2 * n]) + b"\n"
        res["check"] = {"pairs": n, "identical": bool(got == want), "oracle_model_equal": bool(ost["ins_mean"] == model.mean and ost["ins_std_dev"] == model.std_dev)}
    print(json.dumps(res))
    ctx.close(); idx.close()


if __name__ == "__main__":
    main()
