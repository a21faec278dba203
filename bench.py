#!/usr/bin/env python3
"""bench.py — throughput of the HIP hot path on the BASELINE.json workload.

One "step" = one pass of the whole single-end hot path over one resident batch of synthetic 150 bp reads on the
mouse-chr19-scale index built `-r ref -v vcf -H12` style (12 haplotypes that lift onto the reference contig: BASELINE.json
configs[2]): MEM seeding (MS pointers by LF / threshold jumps -> MEMs -> phi / phi^-1 occurrence enumeration, both strands), then
the staged align kernels (chaining, chain selection, lane-per-problem ksw2 DP, traceback, lift-over, MD/NM, MAPQ, the SAM line of
every read) and the host stage that puts the lines in read order.  The reads are in HBM when the timed region starts; the step
ends with the batch's SAM text in host memory.

N > 1: one process per GPU (torch.distributed / RCCL), index replicated, no data-path collective inside the step.
  default       every rank aligns its own --reads reads                                   ("weak":   per-GPU batch fixed)
  --total-reads one read set of that size, sharded by contiguous ranges over the ranks   ("strong": BASELINE.json configs[3])
  --gather-sam  after the timed steps, the per-rank SAM blocks go to rank 0 over RCCL (sizes by all-gather, blocks by
                send/recv) and the time of that gather is reported beside the step time

Prints ONE JSON line (rank 0).  Everything in it is measured in this run: `roofline` prices ms_lf_kernel (the path's HBM-bound
kernel) with HIP events recorded inside the library on the kernel's own stream; `whole_path` prices the step against SURVEY.md
§8(d)'s bytes(read) = 128 S + 64 J + 128 P + C + R with all five counted by the kernels; `align` carries the HIP-event times of
the align kernels by group and the DP rate; `cpu_baseline` is the CPU oracle's whole path on a bounded sample (rank 0, N == 1).
HBM traffic from rocprofv3 PMC passes cannot be collected from inside this process; the last committed pass is named under
`roofline.traffic_static_from` and never mixed into the measured fields.
"""
import argparse
import json
import os
import sys
import time

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); the align stage runs its launches on
# two streams next to a copy stream and two hand-over streams, and with RCCL's own streams in the process (N > 1) some would share
# a queue and serialise.  Read by the HIP runtime when it starts, so it is set before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9      # 256 CUs x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md: a wave64 VALU op issues over 2 cycles)


def log(*a):
    print("[bench %s]" % time.strftime("%H:%M:%S"), *a, file=sys.stderr, flush=True)


def host_cpus() -> int:
    """CPUs this job may use: affinity and cgroup quota (the GPU box gives each job a CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(period))))
    except Exception:
        pass
    return max(1, min(n, 128))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--base-len", type=int, default=61420004)   # GRCm39 chr19
    ap.add_argument("--haps", type=int, default=12)
    ap.add_argument("--reads", type=int, default=1000000, help="reads per GPU (weak scaling)")
    ap.add_argument("--total-reads", type=int, default=0, help="one read set of this size sharded over the ranks (strong scaling)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--repeats", type=float, default=0.0, help="fraction of the base genome made of interspersed repeat copies (SURVEY.md 8(d): 0.05)")
    ap.add_argument("--fasta-index", action="store_true", help="the same text as a FASTA-built index (null lifts)")
    ap.add_argument("--gather-sam", action="store_true", help="N > 1: gather the per-rank SAM blocks on rank 0 over RCCL after the timed steps")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cache", default="/tmp/moni_bench_cache")
    args = ap.parse_args()

    import torch
    from moni_align_amd import dist as mdist
    rank, local_rank, world = mdist.env_world()
    if world != args.gpus:
        log("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # rehearsal overrides (several ranks on one GPU with gloo); the driver's multi-GPU run uses the defaults: RCCL, one GPU per rank
    backend = os.environ.get("MONI_BENCH_BACKEND", "nccl")
    if "MONI_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["MONI_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        dist = mdist.init(backend, rank, world, local_rank)

    from moni_align_amd import capi, index_build, synth

    # ---- inputs (seeded, synthetic: SURVEY.md §8(d)) -------------------------------------------------
    t0 = time.time()
    pg = synth.make_pangenome(args.base_len, args.haps, seed=19, var_seed=12, repeat_frac=args.repeats)
    log("rank %d: pangenome %d sequences, %.1f Mchar in %.1fs" % (rank, len(pg.seqs), sum(len(s) for s in pg.seqs) / 1e6, time.time() - t0))
    os.makedirs(args.cache, exist_ok=True)
    key = "idx_%d_%d_%s_%g.mfi" % (args.base_len, args.haps, "fasta" if args.fasta_index else "lifted", args.repeats)
    path = os.path.join(args.cache, key)
    fi = None
    if rank == 0:
        if os.path.exists(path):
            log("loading cached flat index", path)
            fi = index_build.FlatIndex.load(path)
        else:
            t0 = time.time()
            fi = index_build.build_from_pangenome(pg, device="cuda:%d" % local_rank, log=log, lifted=not args.fasta_index)
            torch.cuda.empty_cache()
            log("flat index built on GPU in %.1fs: n=%d r=%d n/r=%.2f" % (time.time() - t0, fi.n, fi.r, fi.n / fi.r))
            if world > 1 or os.environ.get("MONI_BENCH_SAVE_INDEX"):
                fi.save(path + ".tmp")
                os.replace(path + ".tmp", path)
    if world > 1:
        dist.barrier()
        if rank != 0:
            fi = index_build.FlatIndex.load(path)
    t0 = time.time()
    idx = capi.Index(fi=fi, device=local_rank)
    log("rank %d: device image %.2f GB in %.1fs" % (rank, idx.device_bytes / 1e9, time.time() - t0))
    ctx = capi.Ctx(idx)
    L = args.read_len
    if args.total_reads > 0:          # strong scaling: one read set, this rank's contiguous range of it
        all_reads = synth.make_reads(pg, args.total_reads, L, seed=150)
        lo, hi = mdist.shard_range(args.total_reads, rank, world)
        reads = np.ascontiguousarray(all_reads[lo:hi])
        all_names, all_noff = synth.make_names(args.total_reads)
        names = all_names[int(all_noff[lo]):int(all_noff[hi])]
        noff = (all_noff[lo:hi + 1] - all_noff[lo]).astype(np.uint64)
        del all_reads
        scaling = "strong"
    else:
        reads = synth.make_reads(pg, args.reads, L, seed=150 + rank)
        names, noff = synth.make_names(args.reads)
        scaling = "weak"
    n_mine = reads.shape[0]
    offs = np.arange(0, (n_mine + 1) * L, L, dtype=np.uint64)
    ctx.upload(reads.reshape(-1), offs)
    del pg

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup + timed steps: the whole single-end path over the resident batch ------------------------------
    quals = np.full(n_mine * L, ord("I"), dtype=np.uint8)
    threads = max(1, host_cpus() // max(1, world))          # host stage threads of this rank
    for _ in range(args.warmup):
        ctx.align_run(names, noff, quals, host_threads=threads, want_text=False)
    sync_all()
    kern = np.zeros(7)
    stage = {"seed": 0.0, "align_kernels_span": 0.0, "align_stage": 0.0, "host_stage_busy": 0.0}
    grp = {"chain_plan": 0.0, "dp_lane": 0.0, "select_traceback": 0.0, "finish": 0.0}
    stf = None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sam_len, stf = ctx.align_run(names, noff, quals, host_threads=threads, want_text=False)
        kern += [ctx.kernel_ms(w) if w != 5 else 0.0 for w in range(7)]
        stage["seed"] += stf["t_seed"]; stage["align_kernels_span"] += stf["t_dp_kernel"]; stage["align_stage"] += stf["t_dp"]
        stage["host_stage_busy"] += stf["t_host"]
        grp["chain_plan"] += stf["t_k_chain"]; grp["dp_lane"] += stf["t_k_dp"]; grp["select_traceback"] += stf["t_k_select"]; grp["finish"] += stf["t_k_finish"]
    sync_all()
    elapsed = time.perf_counter() - t0
    elapsed = mdist.max_over_ranks(elapsed, dist, coll_dev)
    kern /= max(1, args.steps)
    for d in (stage, grp):
        for k in d:
            d[k] /= max(1, args.steps)
    cnt = ctx.counters()
    sizes = mdist.gather_counts([stf["aligned"], sam_len, n_mine], dist, coll_dev)     # per-rank record counts

    # the final SAM gather of the north star: per-rank blocks to rank 0 over RCCL, timed on its own
    gather = None
    if args.gather_sam and dist is not None:
        sam_bytes, _ = ctx.align_run(names, noff, quals, host_threads=threads)
        blk = torch.frombuffer(bytearray(sam_bytes), dtype=torch.uint8).to(coll_dev)
        sync_all()
        tg = time.perf_counter()
        got, gsz = mdist.gather_sam(blk, dist, coll_dev)
        sync_all()
        tg = mdist.max_over_ranks(time.perf_counter() - tg, dist, coll_dev)
        gather = {"seconds": tg, "bytes": int(sum(gsz)), "GB/s": sum(gsz) / tg / 1e9 if tg > 0 else None,
                  "note": "SAM blocks device to device into rank 0 (all-gather of sizes + send/recv), outside the timed steps"}
        del got, blk

    # seeding stage alone (BASELINE.json configs[1]), same resident batch
    ts = time.perf_counter()
    n_seed_rep = 3
    for _ in range(n_seed_rep):
        ctx.seed_run(25, True, 1000)
    torch.cuda.synchronize()
    seed_s = (time.perf_counter() - ts) / n_seed_rep
    res = ctx.seed_fetch()
    n_mems, n_occs = len(res["mems"]), len(res["occs"])

    out = None
    if rank == 0:
        S, J, P, C = (int(x) for x in cnt)
        R = int(stf["dp_ref_bytes"])
        n_all = sum(x[2] for x in sizes)
        step_s = elapsed / args.steps
        ms_bytes = 128 * S + 64 * J                 # algorithmic bytes of the LF stage
        ms_s = kern[0] / 1e3
        achieved = ms_bytes / ms_s / 1e9 if ms_s > 0 else 0.0
        layout_bytes = 73 * S      # what the move-structure layout itself needs per LF step: one 64-byte fast row, one 8-byte pointer store, 1/8 of a packed pattern word
        path_bytes = 128 * S + 64 * J + 128 * P + C + R
        value = n_all * args.steps / elapsed
        out = {
            "metric": "aligned reads/s (whole node), %d bp SE, mouse-chr19-scale x%d-haplotype index" % (L, args.haps),
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u64/int32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[%d]: mouse-chr19-scale index (%d bp base%s + %d haplotypes, %s, n=%d, r=%d), "
                                   "%s x %d bp reads resident in HBM -> MEM seeding + staged align kernels (chaining, lane-per-problem ksw2 DP, traceback, "
                                   "lift-over, SAM lines, lines gathered in read order) -> SAM text in pinned host memory, one transfer per sub-batch (%d host threads per GPU stand by for reads handed back)"
                                   % (3 if scaling == "strong" else 2, args.base_len, " with %g interspersed repeats" % args.repeats if args.repeats else "", args.haps,
                                      "FASTA-built: null lifts" if args.fasta_index else "ref+VCF -H12 style: haplotypes lift onto the reference contig", fi.n, fi.r,
                                      ("%d sharded over %d ranks" % (args.total_reads, world)) if scaling == "strong" else ("%d per GPU" % args.reads), L, threads),
                       "reads_per_gpu": n_mine, "read_len": L, "parallelism": "reads sharded x%d, index replicated" % world},
            "roofline": {"bound": "hbm", "kernel": "ms_lf_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "traffic_static_from": "profiles/r02y/pmc_hbm.csv (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of this workload, profiles/run_r02.sh: "
                                                "26.66 GB fetched + 2.40 GB written = 29.06 GB per launch of 1 M reads for 47.18 GB algorithmic, i.e. 0.44 of the HBM peak as "
                                                "counted traffic; counters cannot be read inside this run, so `traffic` stays null)",
                         "algorithmic_bytes_per_launch": ms_bytes, "avg_launch_ms": kern[0], "per_read_bytes": ms_bytes / max(1, n_mine),
                         "layout_model": {"bytes_per_launch": layout_bytes, "GB/s": layout_bytes / ms_s / 1e9 if ms_s > 0 else None,
                                          "frac": layout_bytes / ms_s / 1e9 / HBM_PEAK_GBS if ms_s > 0 else None,
                                          "note": "bytes the move-structure layout needs (one 64-byte fast row per LF step, threshold jumps included): the survey's model credits two requests per step"},
                         "note": "the path's HBM-bound kernel (LF / threshold-jump stage of seeding), one launch per step inside the timed region, HIP events on its own stream"},
            "whole_path": {"bytes_per_read": path_bytes / max(1, n_mine), "formula": "128 S + 64 J + 128 P + C + R (SURVEY.md 8(d)), all counted by the kernels in this run",
                           "S_lf_steps": S, "J_threshold_jumps": J, "P_phi_steps": P, "C_text_bytes": C, "R_dp_target_bytes": R,
                           "GB/s": path_bytes / step_s / 1e9, "frac_of_hbm_peak": path_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                           "note": "rank 0's batch over rank 0's step time: the step as a whole is bound by dependent-access latency in the chaining and record kernels, not by HBM"},
            "align": {"kernels_ms_per_step_summed": {k: v * 1e3 for k, v in grp.items()}, "span_ms_per_step": stage["align_kernels_span"] * 1e3,
                      "note": "HIP-event times of the staged align kernels by group, summed over the sub-batches (two launch streams overlap, so the sum exceeds the span)",
                      "dp_problems": stf["dp_tasks"], "dp_cells": stf["dp_cells"],
                      "dp_gcups_in_kernel": stf["dp_cells"] / grp["dp_lane"] / 1e9 if grp["dp_lane"] > 0 else None,
                      "reads_taken_by_general_kernel": stf["kernel_fallback"], "reads_handed_to_host_pipeline": stf["handed_back"]},
            "stages_s_per_step": stage,
            "aligned_per_step": stf["aligned"], "sam_bytes_per_step": sam_len,
            "aligned_all_ranks": sum(x[0] for x in sizes),
            "kernels_ms": {"ms_lf": kern[0], "mem_count": kern[1], "mem_emit": kern[2], "occ_count": kern[3], "occ_fill": kern[4],
                           "seeding_whole": kern[6]},
            "seeding": {"workload": "BASELINE.json configs[1]: MEM seeding stage alone on the same resident batch",
                        "value": n_all / seed_s, "unit": "reads/s", "ms_per_pass": seed_s * 1e3,
                        "work_per_pass": {"lf_steps": S, "threshold_jumps": J, "phi_steps": P, "text_bytes": C, "mems": n_mems, "occs": n_occs}},
        }
        if gather:
            out["gather"] = gather
        out["host"] = {"cpus_usable": host_cpus(), "cpu_count": os.cpu_count(), "host_threads_per_gpu": threads}
        if world == 1 and not args.no_cpu:
            from oracle import orc as _orc          # the CPU baseline / at-scale checker: the only use of oracle/ in this file
            oidx = _orc.OracleIndex(fi=fi)
            cpu_threads = host_cpus()
            # full path on the CPU (oracle/align.hpp), bounded sample; the same sample is an at-scale SAM identity check
            probe = 2000
            t1 = time.perf_counter()
            _orc.align_batch(oidx, reads[:probe].reshape(-1), offs[:probe + 1], names[:int(noff[probe])], noff[:probe + 1], quals[:probe * L],
                             threads=cpu_threads)
            rate = probe / (time.perf_counter() - t1)
            n_cpu = int(max(probe, min(n_mine, rate * args.cpu_seconds)))
            t1 = time.perf_counter()
            wsam, wc = _orc.align_batch(oidx, reads[:n_cpu].reshape(-1), offs[:n_cpu + 1], names[:int(noff[n_cpu])], noff[:n_cpu + 1],
                                        quals[:n_cpu * L], threads=cpu_threads)
            dtc = time.perf_counter() - t1
            gsam, _ = ctx.align_batch(reads[:n_cpu].reshape(-1), offs[:n_cpu + 1], names[:int(noff[n_cpu])], noff[:n_cpu + 1], quals[:n_cpu * L],
                                      host_threads=threads)
            out["cpu_baseline"] = {"value": n_cpu / dtc, "unit": "reads/s", "cores": cpu_threads, "kind": "port",
                                   "sample": "first %d reads of the same batch, whole SE path, oracle/align.hpp with %d threads" % (n_cpu, cpu_threads),
                                   "sam_identical_on_sample": bool(gsam == wsam)}
            # BASELINE.json configs[0]: the CPU path on one thread (the plumbing / SAM-diff baseline), small sample
            n1 = min(2000, n_cpu)
            t1 = time.perf_counter()
            w1, _ = _orc.align_batch(oidx, reads[:n1].reshape(-1), offs[:n1 + 1], names[:int(noff[n1])], noff[:n1 + 1], quals[:n1 * L], threads=1)
            out["cpu_baseline"]["single_thread"] = {"value": n1 / (time.perf_counter() - t1), "unit": "reads/s", "cores": 1,
                                                    "sample": "first %d reads" % n1, "same_text_as_all_threads": bool(w1 == wsam[:len(w1)])}
            # seeding stage alone on the CPU, same bounded way
            t1 = time.perf_counter()
            oidx.seed_batch(reads[:probe].reshape(-1), offs[:probe + 1], 25, True, 1000, threads=cpu_threads)
            rate = probe / (time.perf_counter() - t1)
            n_cs = int(max(probe, min(n_mine, rate * args.cpu_seconds * 0.5)))
            t1 = time.perf_counter()
            want = oidx.seed_batch(reads[:n_cs].reshape(-1), offs[:n_cs + 1], 25, True, 1000, threads=cpu_threads)
            dt = time.perf_counter() - t1
            k = int(want["read_mem_off"][-1])
            same = (np.array_equal(res["read_mem_off"][:n_cs + 1], want["read_mem_off"]) and
                    np.array_equal(res["mems"]["pos"][:k], want["pos"]) and np.array_equal(res["mems"]["len"][:k].astype(np.uint64), want["len"]) and
                    np.array_equal(res["mems"]["occ_cnt"][:k].astype(np.uint64), want["occ_cnt"]) and
                    np.array_equal(res["occs"][:len(want["occs"])], want["occs"]))
            out["seeding"]["cpu_baseline"] = {"value": n_cs / dt, "unit": "reads/s", "cores": cpu_threads, "kind": "port",
                                              "sample": "first %d reads, same stage, oracle/seed.hpp" % n_cs,
                                              "gpu_matches_cpu_on_sample": bool(same)}
        print(json.dumps(out), flush=True)
    ctx.close()
    idx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
