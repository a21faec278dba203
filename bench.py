#!/usr/bin/env python3
"""bench.py — throughput of the HIP hot path on the BASELINE.json workload.

One "step" = one pass of the whole single-end hot path over one resident batch of synthetic 150 bp reads on
the mouse-chr19-scale x12-haplotype index (BASELINE.json configs[2]): MEM seeding (MS pointers by LF /
threshold jumps -> MEMs -> phi/phi^-1 occurrence enumeration, both strands), then align_kernel (chaining,
chain selection, ksw2 extension DP, MD/NM, MAPQ, the SAM line of every read) and the host stage that puts the lines in read order.
The reads are already in HBM when the timed region starts; the step ends with the batch's SAM text in host
memory.  N > 1: one process per GPU (torch.distributed / RCCL), reads sharded, index replicated, no data-path
collective ("weak" scaling: per-GPU batch fixed).

Prints ONE JSON line (rank 0) with `roofline` (ms_lf_kernel, the path's HBM-bound kernel, HIP-event timed inside
the library on its own stream), `dp` (align_kernel), `seeding` (the seeding stage alone, configs[1]) and
`cpu_baseline` (the CPU oracle's whole path on a bounded sample, rank 0, N == 1 only).
"""
import argparse
import json
import os
import sys
import time

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); the align stage runs its launches on
# two streams next to a copy stream, and with RCCL's own streams in the process (N > 1) two of them could share a queue and
# serialise.  Read by the HIP runtime when it starts, so it is set before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


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
    ap.add_argument("--reads", type=int, default=1000000)       # per GPU
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cache", default="/tmp/moni_bench_cache")
    ap.add_argument("--full-path-reads", type=int, default=0, help="(ignored; the step is the full path)")
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
    pg = synth.make_pangenome(args.base_len, args.haps, seed=19, var_seed=12)
    log("rank %d: pangenome %d sequences, %.1f Mchar in %.1fs" % (rank, len(pg.seqs), sum(len(s) for s in pg.seqs) / 1e6, time.time() - t0))
    os.makedirs(args.cache, exist_ok=True)
    key = "idx_%d_%d.mfi" % (args.base_len, args.haps)
    path = os.path.join(args.cache, key)
    fi = None
    if rank == 0:
        if os.path.exists(path):
            log("loading cached flat index", path)
            fi = index_build.FlatIndex.load(path)
        else:
            t0 = time.time()
            fi = index_build.build_from_pangenome(pg, device="cuda:%d" % local_rank, log=log)
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
    reads = synth.make_reads(pg, args.reads, L, seed=150 + rank)
    offs = np.arange(0, (args.reads + 1) * L, L, dtype=np.uint64)
    ctx.upload(reads.reshape(-1), offs)
    del pg

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup + timed steps: the whole single-end path over the resident batch ------------------------------
    names, noff = synth.make_names(args.reads)
    quals = np.full(args.reads * L, ord("I"), dtype=np.uint8)
    threads = max(1, host_cpus() // max(1, world))          # host stage threads of this rank
    for _ in range(args.warmup):
        ctx.align_run(names, noff, quals, host_threads=threads, want_text=False)
    sync_all()
    kern = np.zeros(7)
    stage = {"seed": 0.0, "align_kernel": 0.0, "align_stage": 0.0, "host_stage_busy": 0.0}
    stf = None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sam_len, stf = ctx.align_run(names, noff, quals, host_threads=threads, want_text=False)
        kern += [ctx.kernel_ms(w) if w != 5 else 0.0 for w in range(7)]
        stage["seed"] += stf["t_seed"]; stage["align_kernel"] += stf["t_dp_kernel"]; stage["align_stage"] += stf["t_dp"]
        stage["host_stage_busy"] += stf["t_host"]
    sync_all()
    elapsed = time.perf_counter() - t0
    elapsed = mdist.max_over_ranks(elapsed, dist, coll_dev)
    kern /= max(1, args.steps)
    for k in stage:
        stage[k] /= max(1, args.steps)
    cnt = ctx.counters()
    sizes = mdist.gather_counts([stf["aligned"], sam_len], dist, coll_dev)     # the only result exchange: per-rank record counts

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
        traffic, traffic_src = None, None
        try:        # HBM bytes per launch from the separate rocprofv3 --pmc passes of this same command (profiles/run_profile.sh)
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_ms_lf.json")))
            if tj.get("n") == fi.n and tj.get("reads") == args.reads and tj.get("read_len") == L:
                traffic, traffic_src = tj["fetch_bytes"] + tj["write_bytes"], tj["source"]
        except Exception:
            pass
        sq = None
        try:        # issue counters of align_kernel from the SQ pass of the same recipe
            sq = json.load(open(os.path.join(ROOT, "profiles", "align_kernel_sq.json")))
        except Exception:
            pass
        ms_bytes = 128 * S + 64 * J                 # algorithmic bytes of the LF stage
        ms_s = kern[0] / 1e3
        achieved = ms_bytes / ms_s / 1e9 if ms_s > 0 else 0.0
        value = world * args.reads * args.steps / elapsed
        out = {
            "metric": "aligned reads/s (whole node), %d bp SE, mouse-chr19-scale x%d-haplotype index" % (L, args.haps),
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64/int32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: mouse-chr19-scale index (%d bp base + %d haplotypes, n=%d, r=%d), "
                                   "%d x %d bp reads per GPU resident in HBM -> MEM seeding + HIP ksw2 extension (align_kernel) -> "
                                   "SAM text (lines spelled in align_kernel, put in read order by %d host threads per GPU, overlapped)"
                                   % (args.base_len, args.haps, fi.n, fi.r, args.reads, L, threads),
                       "reads_per_gpu": args.reads, "read_len": L, "parallelism": "reads sharded x%d, index replicated" % world},
            "roofline": {"bound": "hbm", "kernel": "ms_lf_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": ms_bytes, "avg_launch_ms": kern[0],
                         "per_read_bytes": ms_bytes / args.reads,
                         "note": "the path's HBM-bound kernel (LF / threshold-jump stage of seeding), one launch per step inside the timed "
                                 "region; the step's longest kernel, align_kernel, is bound by dependent-access latency and the DP's integer VALU chain: see dp"},
            "dp": {"kernel": "align_kernel", "bound": "latency + valu-int32 (no HBM or MFMA roofline applies)", "launches_per_step": stf["dp_rounds"],
                   "ms_per_step": stage["align_kernel"] * 1e3, "dp_problems": stf["dp_tasks"], "dp_cells": stf["dp_cells"],
                   "gcups": stf["dp_cells"] / stage["align_kernel"] / 1e9 if stage["align_kernel"] > 0 else None,
                   "handed_back_to_host_pipeline": stf["handed_back"],
                   "dp_problems_reused_from_memo": stf["dp_reused"], "dp_cells_reused": stf["dp_cells_reused"]},
            "stages_s_per_step": stage,
            "aligned_per_step": stf["aligned"], "sam_bytes_per_step": sam_len,
            "aligned_all_ranks": sum(x[0] for x in sizes),
            "kernels_ms": {"ms_lf": kern[0], "mem_count": kern[1], "mem_emit": kern[2], "occ_count": kern[3], "occ_fill": kern[4],
                           "seeding_whole": kern[6]},
            "seeding": {"workload": "BASELINE.json configs[1]: MEM seeding stage alone on the same resident batch",
                        "value": world * args.reads / seed_s, "unit": "reads/s", "ms_per_pass": seed_s * 1e3,
                        "work_per_pass": {"lf_steps": S, "threshold_jumps": J, "phi_steps": P, "text_bytes": C, "mems": n_mems, "occs": n_occs}},
        }
        if sq and stage["align_kernel"] > 0 and traffic is not None:        # same workload as the profiled one (traffic matched n / reads / read_len)
            # integer-VALU view of align_kernel: wave-instructions counted by rocprofv3 (per launch of sq["reads_per_launch"] reads),
            # priced against 256 CUs x 4 SIMDs x 32 lanes/cycle (MI355X_MICROARCH.md: a wave64 VALU op retires in 2 cycles) at 2.4 GHz
            insts = sq["valu_wave_insts_per_launch"] * args.reads / sq["reads_per_launch"]
            peak = 256 * 4 * 32 * 2.4e9
            out["dp"]["valu"] = {"wave_insts_per_step": insts, "achieved_lane_ops_per_s": insts * 64 / stage["align_kernel"], "peak_lane_ops_per_s": peak,
                                 "frac": insts * 64 / stage["align_kernel"] / peak,
                                 "waves_parked_frac": sq["sq_wait_any_quad"] / sq["sq_wave_cycles_quad"],
                                 "hbm_bytes_per_step": (sq["fetch_bytes_per_launch"] + sq["write_bytes_per_launch"]) * args.reads / sq["reads_per_launch"],
                                 "source": sq["source"]}
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
            n_cpu = int(max(probe, min(args.reads, rate * args.cpu_seconds)))
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
                                                    "sample": "first %d reads" % n1, "same_text_as_16_threads": bool(w1 == wsam[:len(w1)])}
            # seeding stage alone on the CPU, same bounded way
            t1 = time.perf_counter()
            oidx.seed_batch(reads[:probe].reshape(-1), offs[:probe + 1], 25, True, 1000, threads=cpu_threads)
            rate = probe / (time.perf_counter() - t1)
            n_cs = int(max(probe, min(args.reads, rate * args.cpu_seconds * 0.5)))
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
