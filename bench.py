#!/usr/bin/env python3
"""bench.py — throughput of the HIP hot path on the BASELINE.json workload.

One "step" = one pass of the whole single-end hot path over this rank's reads, resident in HBM, on the mouse-chr19-scale index
built `-r ref -v vcf -H12` style (12 haplotypes that lift onto the reference contig): MEM seeding (MS pointers by LF / threshold
jumps -> MEMs -> phi / phi^-1 occurrence enumeration, both strands), then the staged align kernels (chaining, chain selection,
lane-per-problem ksw2 DP, traceback, lift-over, MD/NM, MAPQ, the SAM line of every read, lines put in read order).  The step ends
with the SAM text in pinned host memory.

  N = 1 (default)   BASELINE.json configs[2]: 1 M x 150 bp reads, one resident batch                          ("weak")
  N > 1 (default)   BASELINE.json configs[3]: ONE read set of 10 M reads, sharded by contiguous ranges over the ranks; a rank
                    goes through its shard in resident chunks of <= --reads reads (moni_reads_swap)           ("strong")
                    and, after the timed steps, the per-rank SAM blocks go to rank 0 over RCCL (sizes by all-gather, blocks by
                    send/recv): the `gather` block of the line; --no-gather-sam skips it
  --total-reads T   the sharded mode with T reads at any N (N = 1: the same set on one GPU, the base of the scaling curve)

`python bench.py --gpus N` starts the N rank processes itself when it is not already running under a launcher (no RANK in the
environment): fresh children, one GPU each, RCCL; the parent never touches the GPU.  Under `python -m torch.distributed.run` the
ranks come from the launcher.  --dry-run rehearses the launch, the sharding and the gather on CPU (gloo) without any GPU.

Prints ONE JSON line (rank 0).  Everything in it is measured in this run: `roofline` prices ms_lf_kernel (the path's HBM-bound
kernel) with HIP events recorded inside the library on the kernel's own stream; `whole_path` prices the step against SURVEY.md
§8(d)'s bytes(read) = 128 S + 64 J + 128 P + C + R with all five counted by the kernels; `align` carries the HIP-event times of
the align kernels by group and the DP rate; `from_host` is the same batch from host memory to SAM text in host memory
(moni_align_stream: upload included); `cpu_baseline` is the CPU oracle's whole path on a bounded sample (rank 0, N == 1).
HBM traffic from rocprofv3 PMC passes cannot be collected from inside this process; the last committed pass is named under
`roofline.traffic_static_from` and never mixed into the measured fields.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); the align stage runs its launches on
# two streams next to a copy stream and two hand-over streams (the paired path on ~11), and with RCCL's own streams in the process
# (N > 1) some would share a queue and serialise (profiles/r03o: 4 queues 258 ms, 8: 214, 16: 194 per 1 M pairs).  Read by the HIP
# runtime when it starts, so it is set before torch is imported; the library sets the same default (moni_hip_default_queues).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# The image's environment exports this (dmabuf IPC: the host driver supports no other, and RCCL between processes fails with
# hipIpcGetMemHandle: invalid argument without it); kept for a launcher that starts the ranks from a clean environment.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
CONFIGS3_READS = 10_000_000


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--base-len", type=int, default=61420004)   # GRCm39 chr19
    ap.add_argument("--haps", type=int, default=12)
    ap.add_argument("--reads", type=int, default=1000000, help="reads of one resident batch (N = 1: the batch; sharded mode: the largest chunk)")
    ap.add_argument("--total-reads", type=int, default=-1,
                    help="one read set of this size sharded over the ranks (strong scaling); default: %d when N > 1, off when N = 1; 0 = off (every rank its own --reads reads: weak)" % CONFIGS3_READS)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--repeats", type=float, default=0.0, help="fraction of the base genome made of interspersed repeat copies (SURVEY.md 8(d): 0.05)")
    ap.add_argument("--fasta-index", action="store_true", help="the same text as a FASTA-built index (null lifts)")
    ap.add_argument("--gather-sam", dest="gather_sam", action="store_true", default=None, help="gather the per-rank SAM blocks on rank 0 over RCCL after the timed steps (default at N > 1)")
    ap.add_argument("--no-gather-sam", dest="gather_sam", action="store_false")
    ap.add_argument("--verify-gather", action="store_true", help="rank 0 also aligns the whole read set by itself and compares the gathered text with it byte for byte (rehearsals: use a small --total-reads)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-from-host", action="store_true")
    ap.add_argument("--inflight", type=int, default=2, help="contexts of this rank that work through the timed steps side by side, each with the batch resident (one caller thread per context, as moni-hip-align's workers): "
                                                            "the seeding kernels of one step overlap the align kernels of another; every step is still one whole pass over the batch")
    ap.add_argument("--no-single-context", action="store_true", help="skip the leg that runs a few steps with one context alone after the timed region (--inflight > 1)")
    ap.add_argument("--n-rate", type=float, default=0.0, help="robustness leg: this fraction of the reads gets one N at a random place (real Illumina data: 0.5-2 %% of the reads hold an N); the DP problems that touch it leave the packed 2-bit kernels")
    ap.add_argument("--no-scaling-base", action="store_true", help="N = 1: skip the extra leg that runs configs[3]'s read set (--scaling-base-reads reads, resident chunks of --reads) on the one GPU")
    ap.add_argument("--scaling-base-reads", type=int, default=CONFIGS3_READS)
    ap.add_argument("--paired", action="store_true", help="the paired-end path instead (moni_pe_learn_batch / moni_pe_align_batch over --pairs FR pairs of 2 x --read-len, orphan recovery on): pairs/s")
    ap.add_argument("-Z", "--secondary-chains", dest="secondary_chains", action="store_true", help="--paired: find_chains_secondary (the reference's -Z) in the learning batches and the alignment")
    ap.add_argument("--pairs", type=int, default=1000000, help="--paired: read pairs (per GPU; sharded by contiguous ranges like the reads when --total-reads is given)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: launch, rendezvous (gloo), sharding and the SAM gather with placeholder records")
    ap.add_argument("--cache", default="/tmp/moni_bench_cache")
    return ap.parse_args(argv)


# ---- launcher: `python bench.py --gpus N` outside torch.distributed.run -------------------------------------------------
def launch_ranks(args) -> int:
    """Start args.gpus rank processes (fresh interpreters; this parent has not initialised HIP and never does), one GPU each, and
    wait for them.  Rank 0 inherits stdout (the JSON line); the other ranks' stdout goes to stderr."""
    n = args.gpus
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "MONI_BENCH_SELF_LAUNCHED": "1"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    log("launcher: started %d ranks (pids %s), rendezvous 127.0.0.1:%d" % (n, [p.pid for p in procs], port))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                log("launcher: rank %d exited with code %d; stopping the others" % (r, code))
                for o in alive:
                    procs[o].terminate()          # exactly the children started above
        time.sleep(0.2)
    return rc


# ---- dry run: the multi-rank plumbing without a GPU -----------------------------------------------------------------------
def placeholder_block(lo: int, hi: int) -> bytes:
    """unaligned SAM records (flag 4) of reads [lo, hi): stands in for a rank's SAM block in the dry run"""
    return b"".join(b"simulated.%d\t4\t*\t0\t255\t*\t*\t0\t0\tACGT\tIIII\n" % i for i in range(lo, hi))


def dry_run(args) -> int:
    import torch      # noqa: F401
    from moni_align_amd import dist as mdist
    rank, local_rank, world = mdist.env_world()
    dist = mdist.init("gloo", rank, world) if world > 1 else None
    total = args.total_reads if args.total_reads > 0 else (CONFIGS3_READS if world > 1 else args.reads)
    total = min(total, 200000)                       # placeholder text only
    lo, hi = mdist.shard_range(total, rank, world)
    chunks = chunk_bounds(hi - lo, args.reads)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    blk = placeholder_block(lo, hi)
    elapsed = mdist.max_over_ranks(time.perf_counter() - t0, dist)
    sizes = mdist.gather_counts([hi - lo, len(blk), len(chunks) - 1], dist)
    tg = time.perf_counter()
    got, gsz = mdist.gather_sam(blk, dist)
    tg = mdist.max_over_ranks(time.perf_counter() - tg, dist)
    if rank == 0:
        same = bytes(got.numpy().tobytes()) == placeholder_block(0, total)
        print(json.dumps({"metric": "dry run (no GPU): launch, rendezvous, sharding, gather", "dry_run": True, "value": 0.0, "unit": "reads/s",
                          "n_gpus": world, "steps": 0, "warmup": 0, "ms_per_step": elapsed * 1e3, "higher_is_better": True,
                          "scaling": "strong" if world > 1 or args.total_reads > 0 else "weak", "vs_baseline": None, "data": "placeholder records",
                          "config": {"workload": "placeholder", "total_reads": total, "reads_per_rank": [x[0] for x in sizes], "chunks_per_rank": [x[2] for x in sizes]},
                          "gather": {"seconds": tg, "bytes": int(sum(gsz)), "per_rank_bytes": gsz, "identical_to_unsharded": bool(same)},
                          "launched_by": "bench.py" if os.environ.get("MONI_BENCH_SELF_LAUNCHED") else "external launcher"}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def chunk_bounds(n: int, chunk_max: int, multiple_of: int = 1):
    """even split of n reads into the fewest chunks of at most chunk_max reads - when more than one is needed, a multiple of `multiple_of` of them (the contexts
    of a rank take turns at the chunks: an equal share each): k+1 boundaries"""
    k = max(1, -(-n // max(1, chunk_max)))
    if k > 1 and multiple_of > 1:
        k = -(-k // multiple_of) * multiple_of
    return [n * i // k for i in range(k + 1)]


# ---- one rank ---------------------------------------------------------------------------------------------------------------
def run_rank(args) -> int:
    import torch
    from moni_align_amd import dist as mdist
    rank, local_rank, world = mdist.env_world()
    if world != args.gpus:
        log("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback (--dry-run rehearses the multi-rank plumbing without one)")
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

    total = args.total_reads if args.total_reads >= 0 else (CONFIGS3_READS if world > 1 else 0)
    sharded = total > 0
    gather_on = (world > 1) if args.gather_sam is None else (args.gather_sam and world > 1)

    # ---- inputs (seeded, synthetic: SURVEY.md §8(d)) -------------------------------------------------
    # Wall time of every phase in front of the timed region goes to stderr (rank 0) and into the line (`setup_s`): at N ranks the order is
    # pangenome (every rank, side by side) -> rank 0 builds the flat index on its GPU and writes the cache file while the others generate their reads
    # and drop the pangenome -> barrier -> every rank maps the file (one copy in the page cache for all) and builds its device image, side by side.
    t_start = time.time()
    phases = []

    def phase(name, t_from):
        phases.append((name, time.time() - t_from))
        log("rank %d: %s %.1fs (at %.1fs)" % (rank, name, phases[-1][1], time.time() - t_start))

    t0 = time.time()
    pg = synth.make_pangenome(args.base_len, args.haps, seed=19, var_seed=12, repeat_frac=args.repeats)
    n_seqs, n_chars = len(pg.seqs), sum(len(x) for x in pg.seqs)
    phase("pangenome (%d sequences, %.1f Mchar)" % (n_seqs, n_chars / 1e6), t0)
    os.makedirs(args.cache, exist_ok=True)
    key = "idx_%d_%d_%s_%g.mfi" % (args.base_len, args.haps, "fasta" if args.fasta_index else "lifted", args.repeats)
    path = os.path.join(args.cache, key)
    L = args.read_len

    def my_reads():          # this rank's reads (host arrays); the pangenome is not needed after them
        if args.paired:
            return None
        if sharded:            # strong scaling: one read set, this rank's contiguous range of it, in resident chunks
            lo_, hi_ = mdist.shard_range(total, rank, world)
            rd = synth.make_reads_range(pg, lo_, hi_, L, seed=1500, threads=max(1, host_cpus() // max(1, world)))
            nm_, no_ = synth.make_names_range(lo_, hi_)
            return lo_, hi_, rd, nm_, no_
        rd = synth.make_reads(pg, args.reads, L, seed=150 + rank)
        nm_, no_ = synth.make_names(args.reads)
        return 0, args.reads, rd, nm_, no_

    fi = None
    mine = None
    if rank == 0:
        if os.path.exists(path):
            t0 = time.time()
            fi = index_build.FlatIndex.load(path, mmap=True)
            phase("cached flat index mapped (%s)" % path, t0)
        else:
            t0 = time.time()
            fi = index_build.build_from_pangenome(pg, device="cuda:%d" % local_rank, log=log, lifted=not args.fasta_index)
            torch.cuda.empty_cache()
            phase("flat index built on the GPU (n=%d r=%d n/r=%.2f)" % (fi.n, fi.r, fi.n / fi.r), t0)
            if world > 1 or os.environ.get("MONI_BENCH_SAVE_INDEX"):
                t0 = time.time()
                try:
                    fi.save(path + ".tmp")
                    os.replace(path + ".tmp", path)
                    phase("index cache written", t0)
                except OSError as e:          # (no room for the 10 GB cache file: the other ranks then build the index on their own GPUs)
                    log("could not write the index cache %s: %s" % (path, e))
                    try:
                        os.remove(path + ".tmp")
                    except OSError:
                        pass
    elif not args.paired:
        t0 = time.time()
        mine = my_reads()          # while rank 0 builds the index
        pg = None                  # (0.8 GB; not needed again unless the cache file turns out to be unusable)
        phase("reads generated, pangenome dropped", t0)
    if world > 1:
        t0 = time.time()
        dist.barrier()
        phase("barrier behind rank 0's index", t0)
    t0 = time.time()
    if rank == 0:
        idx = capi.Index(fi=fi, device=local_rank)
    else:
        try:
            idx = capi.Index(path=path, device=local_rank)          # the file mapped by the library: no private host copy
        except Exception as e:
            log("rank %d: no usable index cache (%s): building on cuda:%d" % (rank, e, local_rank))
            if pg is None:
                pg = synth.make_pangenome(args.base_len, args.haps, seed=19, var_seed=12, repeat_frac=args.repeats)
            fi = index_build.build_from_pangenome(pg, device="cuda:%d" % local_rank, log=log, lifted=not args.fasta_index)
            torch.cuda.empty_cache()
            idx = capi.Index(fi=fi, device=local_rank)
    fi_n, fi_r = idx.n, idx.r
    phase("device image (%.2f GB)" % (idx.device_bytes / 1e9), t0)
    ctx = capi.Ctx(idx)
    if args.paired:
        pass
    else:
        if mine is None:
            t0 = time.time()
            mine = my_reads()
            phase("reads generated", t0)
        lo, hi, reads, names, noff = mine
        scaling = "strong" if sharded else "weak"
        if args.n_rate > 0:          # one N in a seeded choice of the reads (by global read number: the same reads whatever the sharding)
            rn = np.random.Generator(np.random.MT19937(4242))
            pick = rn.random(max(hi, 1))[lo:hi] < args.n_rate
            at = rn.integers(0, L, size=max(hi, 1))[lo:hi]
            reads[np.nonzero(pick)[0], at[pick]] = ord("N")
    if args.paired:
        return run_paired(args, rank, world, dist, coll_dev, backend, pg, fi, idx, ctx, phases, t_start)
    n_mine = reads.shape[0]
    cb = chunk_bounds(n_mine, args.reads, max(1, args.inflight))
    n_chunks = len(cb) - 1
    quals = np.full(n_mine * L, ord("I"), dtype=np.uint8)
    # contexts of this rank (--inflight): with ONE resident chunk every context holds the same batch and they share the timed steps; with several chunks
    # (a sharded read set) chunk k belongs to context k % inflight, parked there (moni_reads_swap) and swapped in for its turn - a step is still every chunk once
    inflight = max(1, args.inflight)
    ctxs = [ctx] + [capi.Ctx(idx) for _ in range(inflight - 1)]
    shared_batch = n_chunks == 1
    owner = [0 if shared_batch else k % inflight for k in range(n_chunks)]
    slot = [0 if shared_batch else k // inflight for k in range(n_chunks)]
    n_own = [sum(1 for o in owner if o == i) for i in range(inflight)]
    chunk = []          # per chunk: (names, name offsets, quals) views + the read count
    for k in range(n_chunks):
        a, b = cb[k], cb[k + 1]
        offs_k = np.arange(0, (b - a + 1) * L, L, dtype=np.uint64)
        for cx in (ctxs if shared_batch else [ctxs[owner[k]]]):
            cx.upload(reads[a:b].reshape(-1), offs_k)
        if n_own[owner[k]] > 1:
            ctxs[owner[k]].swap(slot[k])                  # parked in HBM; swapped in for its turn
        chunk.append((names[int(noff[a]):int(noff[b])], (noff[a:b + 1] - noff[a]).astype(np.uint64), quals[a * L:b * L], b - a))
    offs = np.arange(0, (n_mine + 1) * L, L, dtype=np.uint64)
    pg_keep = pg if (rank == 0 and world == 1 and not sharded and not args.no_scaling_base) else None      # (the scaling-base leg generates its read set from it)
    del pg
    phases.append(("reads [%d, %d) resident in %d chunk(s)" % (lo, hi, n_chunks), 0.0))
    log("rank %d: reads [%d, %d) resident in %d chunk(s); setup took %.1fs" % (rank, lo, hi, n_chunks, time.time() - t_start))
    setup_s = time.time() - t_start

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    threads = max(1, host_cpus() // max(1, world))          # host stage threads of this rank

    threads_step = max(1, threads // inflight)
    if inflight > 1:          # two steps side by side fill the GPU by themselves: larger sub-batches (fewer, longer launches) do better then (profiles/r04o)
        os.environ.setdefault("MONI_ALIGN_SUB", "500000")

    def one_pass(want_text=False, acc=None, ci=None):
        """the whole path over the resident chunks (ci: those of context ci; None: all of this rank's, in order); returns (SAM bytes or total length, stats of the last chunk)"""
        outs, st_last, tot_len = [], None, 0
        for k in range(n_chunks):
            if ci is not None and not shared_batch and owner[k] != ci:
                continue
            cx = ctxs[ci if (shared_batch and ci is not None) else owner[k]]
            nm, no, ql, _ = chunk[k]
            parked = n_own[owner[k]] > 1
            if parked:
                cx.swap(slot[k])
            sam, st = cx.align_run(nm, no, ql, host_threads=threads if inflight == 1 else threads_step, want_text=want_text)
            if acc is not None:
                acc(st, cx)
            if parked:
                cx.swap(slot[k])
            if want_text:
                outs.append(sam)
            else:
                tot_len += sam
            st_last = st
        return (b"".join(outs) if want_text else tot_len), st_last

    # ---- warmup + timed steps ------------------------------------------------------------------------------------------------
    for _ in range(args.warmup):
        for ci in range(inflight):
            one_pass(ci=ci)
    sync_all()
    kern = np.zeros(7)
    stage = {"seed": 0.0, "align_kernels_span": 0.0, "align_stage": 0.0, "host_stage_busy": 0.0}
    grp = {"chain_plan": 0.0, "dp_lane": 0.0, "select_traceback": 0.0, "finish": 0.0}
    tot = {"aligned": 0, "dp_tasks": 0, "dp_cells": 0, "kernel_fallback": 0, "handed_back": 0, "dp_ref_bytes": 0, "dp_cells_cut": 0, "dp_slots": 0}
    cnt = np.zeros(4, dtype=np.uint64)
    n_calls = [0]

    why = {}

    acc_lock = threading.Lock()

    def acc(st, cx):
      with acc_lock:
        for k2, v in st["handover_why"].items():
            why[k2] = why.get(k2, 0) + v
        kern[:] += [cx.kernel_ms(w) if w != 5 else 0.0 for w in range(7)]
        stage["seed"] += st["t_seed"]; stage["align_kernels_span"] += st["t_dp_kernel"]; stage["align_stage"] += st["t_dp"]
        stage["host_stage_busy"] += st["t_host"]
        grp["chain_plan"] += st["t_k_chain"]; grp["dp_lane"] += st["t_k_dp"]; grp["select_traceback"] += st["t_k_select"]; grp["finish"] += st["t_k_finish"]
        for k2 in tot:
            tot[k2] += st[k2]
        cnt[:] += cx.counters()
        n_calls[0] += 1

    sam_len = 0
    t0 = time.perf_counter()
    if inflight == 1:
        for _ in range(args.steps):
            sam_len, _ = one_pass(acc=acc)
    elif shared_batch:          # exactly args.steps passes in all: a context takes the next one as soon as it is through with its last
        left = [args.steps]
        lens = []

        def stepper(ci):
            while True:
                with acc_lock:
                    if left[0] <= 0:
                        return
                    left[0] -= 1
                n, _ = one_pass(acc=acc, ci=ci)
                lens.append(n)
        th = [threading.Thread(target=stepper, args=(ci,)) for ci in range(inflight)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        sam_len = lens[-1] if lens else 0
    else:                       # every context goes args.steps times through its own chunks; the contexts side by side
        lens = [0] * inflight

        def stepper(ci):
            for _ in range(args.steps):
                lens[ci], _ = one_pass(acc=acc, ci=ci)
        th = [threading.Thread(target=stepper, args=(ci,)) for ci in range(inflight)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        sam_len = sum(lens)
    sync_all()
    elapsed = time.perf_counter() - t0
    elapsed = mdist.max_over_ranks(elapsed, dist, coll_dev)
    steps = max(1, args.steps)
    kern_launch = kern / max(1, n_calls[0])          # per launch (= per chunk)
    for d in (stage, grp):
        for k in d:
            d[k] /= steps                             # per step (all chunks of the rank)
    for k in tot:
        tot[k] //= steps
    cnt = cnt // np.uint64(steps)
    sizes = mdist.gather_counts([tot["aligned"], sam_len, n_mine], dist, coll_dev)     # per-rank record counts
    mem_free, mem_total = torch.cuda.mem_get_info()          # with every context's working memory allocated (grow-only buffers, kept across batches)
    # one context alone (no second step beside it): the latency of a step, and the kernels' durations without a neighbour
    single = None
    if inflight > 1 and shared_batch and not args.no_single_context:
        sub_was = os.environ.pop("MONI_ALIGN_SUB", None)
        one_pass(ci=0)
        sync_all()
        n1s = max(1, min(4, args.steps))
        t1 = time.perf_counter()
        for _ in range(n1s):
            _, st1 = ctx.align_run(chunk[0][0], chunk[0][1], chunk[0][2], host_threads=threads, want_text=False)
        sync_all()
        e1 = mdist.max_over_ranks(time.perf_counter() - t1, dist, coll_dev)
        single = {"ms_per_step": e1 / n1s * 1e3, "steps": n1s, "reads_per_s_this_rank": chunk[0][3] * n1s / e1, "ms_lf_kernel_ms": ctx.kernel_ms(0),
                  "stages_ms": {"seed": st1["t_seed"] * 1e3, "align_kernels_span": st1["t_dp_kernel"] * 1e3},
                  "note": "the same resident batch, one context by itself (sub-batches of 250 000), after the timed region"}
        if sub_was is not None:
            os.environ["MONI_ALIGN_SUB"] = sub_was

    # ---- the final SAM gather of the north star: per-rank blocks to rank 0 over RCCL, timed on its own ----------------------
    gather = None
    if gather_on:
        sam_bytes, _ = one_pass(want_text=True)
        blk = torch.frombuffer(bytearray(sam_bytes), dtype=torch.uint8).to(coll_dev)
        del sam_bytes
        sync_all()
        tg = time.perf_counter()
        got, gsz = mdist.gather_sam(blk, dist, coll_dev)
        sync_all()
        tg = mdist.max_over_ranks(time.perf_counter() - tg, dist, coll_dev)
        gather = {"seconds": tg, "bytes": int(sum(gsz)), "GB/s": sum(gsz) / tg / 1e9 if tg > 0 else None, "per_rank_bytes": gsz, "backend": backend,
                  "note": "SAM blocks device to device into rank 0, in rank (= input) order (all-gather of sizes + send/recv), after the timed steps"}
        if rank == 0:
            n_lines = int((got == 10).sum().item())
            gather["records"] = n_lines
            gather["records_match_reads"] = bool(n_lines == sum(x[2] for x in sizes))
            if args.verify_gather and sharded:
                # the whole set on this rank alone, chunk by chunk in a second context: the unsharded text
                pg2 = synth.make_pangenome(args.base_len, args.haps, seed=19, var_seed=12, repeat_frac=args.repeats)
                ctx2 = capi.Ctx(idx)
                whole = []
                wb = chunk_bounds(total, args.reads)
                for k in range(len(wb) - 1):
                    rk = synth.make_reads_range(pg2, wb[k], wb[k + 1], L, seed=1500)
                    nk, nok = synth.make_names_range(wb[k], wb[k + 1])
                    ok = np.arange(0, (rk.shape[0] + 1) * L, L, dtype=np.uint64)
                    s, _ = ctx2.align_batch(rk.reshape(-1), ok, nk, nok, np.full(rk.size, ord("I"), np.uint8), host_threads=threads, stream=True)
                    whole.append(s)
                ctx2.close()
                del pg2
                gather["identical_to_unsharded"] = bool(bytes(got.cpu().numpy().tobytes()) == b"".join(whole))
        del got, blk
    for cx in ctxs[1:]:
        cx.close()

    # ---- seeding stage alone (BASELINE.json configs[1]) on the first resident chunk ------------------------------------------
    if n_own[0] > 1:
        ctx.swap(0)
    ts = time.perf_counter()
    n_seed_rep = 3
    for _ in range(n_seed_rep):
        ctx.seed_run(25, True, 1000)
    torch.cuda.synchronize()
    seed_s = (time.perf_counter() - ts) / n_seed_rep
    res = ctx.seed_fetch()
    n_mems, n_occs = len(res["mems"]), len(res["occs"])
    n_first = chunk[0][3]

    # ---- the same batch from host memory: upload + whole path + text in host memory (moni_align_stream) ----------------------
    from_host = None
    if rank == 0 and world == 1 and not args.no_from_host:
        nm, no, ql, nb = chunk[0]
        rd = reads[:nb].reshape(-1)
        ob = offs[:nb + 1]
        ctx.align_batch(rd, ob, nm, no, ql, host_threads=threads, stream=True, want_text=False)
        reps = max(3, min(10, args.steps))
        t1 = time.perf_counter()
        for _ in range(reps):
            ctx.align_batch(rd, ob, nm, no, ql, host_threads=threads, stream=True, want_text=False)
        one_s = (time.perf_counter() - t1) / reps
        # a streaming caller keeps two contexts per GPU going (moni-hip-align runs three): one uploads / seeds while the other aligns
        ctx_b = capi.Ctx(idx)
        ctx_b.align_batch(rd, ob, nm, no, ql, host_threads=threads, stream=True, want_text=False)

        def worker(cx, n):
            for _ in range(n):
                cx.align_batch(rd, ob, nm, no, ql, host_threads=max(1, threads // 2), stream=True, want_text=False)
        th = [threading.Thread(target=worker, args=(cx, reps)) for cx in (ctx, ctx_b)]
        t1 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        two_s = (time.perf_counter() - t1) / (2 * reps)
        ctx_b.close()
        from_host = {"value": nb / one_s, "unit": "reads/s", "ms_per_batch": one_s * 1e3, "batch_reads": nb,
                     "two_contexts": {"value": nb / two_s, "unit": "reads/s", "ms_per_batch": two_s * 1e3,
                                      "note": "two contexts on the GPU, one caller thread each (the arrangement of moni-hip-align's workers): uploads and seeding of one overlap the align kernels of the other"},
                     "note": "moni_align_stream: reads, names, qualities in pageable host memory -> SAM text in the context's pinned buffer; upload inside the timed call (align_full_ksw2.cpp:333,399-402 wraps file -> file; FASTQ parsing and file writes are the front end's, profiles/frontend.py)"}

    # ---- N = 1: configs[3]'s read set on the one GPU (the base of the 1 -> N curve in the shape the N > 1 runs have) ------------------------
    scaling_base = None
    if pg_keep is not None:
        tb0 = time.time()
        sb_total = max(args.reads, args.scaling_base_reads)
        sb_reads = synth.make_reads_range(pg_keep, 0, sb_total, L, seed=1500, threads=host_cpus())
        sb_names, sb_noff = synth.make_names_range(0, sb_total)
        t_gen = time.time() - tb0
        del pg_keep
        sb_ctx = [capi.Ctx(idx) for _ in range(inflight)]          # chunk k on context k % inflight, as the sharded runs hold theirs
        sb_cb = chunk_bounds(sb_total, args.reads, inflight)
        sb_n = len(sb_cb) - 1
        sb_q = np.full(args.reads * L + L, ord("I"), dtype=np.uint8)
        sb_own = [sum(1 for k in range(sb_n) if k % inflight == i) for i in range(inflight)]
        for k in range(sb_n):
            a, b = sb_cb[k], sb_cb[k + 1]
            cx = sb_ctx[k % inflight]
            cx.upload(sb_reads[a:b].reshape(-1), np.arange(0, (b - a + 1) * L, L, dtype=np.uint64))
            if sb_own[k % inflight] > 1:
                cx.swap(k // inflight)
        sb_al_of = [0] * inflight

        def sb_ctx_pass(i):
            al = 0
            for k in range(i, sb_n, inflight):
                a, b = sb_cb[k], sb_cb[k + 1]
                cx = sb_ctx[i]
                if sb_own[i] > 1:
                    cx.swap(k // inflight)
                _, st_k = cx.align_run(sb_names[int(sb_noff[a]):int(sb_noff[b])], (sb_noff[a:b + 1] - sb_noff[a]).astype(np.uint64), sb_q[:(b - a) * L], host_threads=threads if inflight == 1 else threads_step, want_text=False)
                if sb_own[i] > 1:
                    cx.swap(k // inflight)
                al += st_k["aligned"]
            sb_al_of[i] = al

        def sb_pass():
            th = [threading.Thread(target=sb_ctx_pass, args=(i,)) for i in range(inflight)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            return sum(sb_al_of)
        sb_pass()
        torch.cuda.synchronize()
        sb_steps = 2
        t1 = time.perf_counter()
        for _ in range(sb_steps):
            sb_al = sb_pass()
        torch.cuda.synchronize()
        sb_s = (time.perf_counter() - t1) / sb_steps
        for cx in sb_ctx:
            cx.close()
        del sb_reads
        scaling_base = {"workload": "BASELINE.json configs[3]'s read set on this one GPU: %d reads in %d resident chunks of <= %d (what `--total-reads %d` runs at N = 1; the N > 1 default shards the same set)"
                                    % (sb_total, sb_n, args.reads, sb_total),
                        "contexts_in_flight": inflight,
                        "value": sb_al / sb_s, "unit": "aligned reads/s", "reads_per_s": sb_total / sb_s, "ms_per_step": sb_s * 1e3, "steps": sb_steps, "warmup": 1, "scaling": "strong",
                        "reads_generated_s": t_gen, "leg_s": time.time() - tb0}
        log("scaling base: %d reads in %d chunks, %.1f ms per pass" % (sb_total, sb_n, sb_s * 1e3))

    if rank == 0:
        S, J, P, C = (int(x) for x in cnt)          # per step, this rank
        R = int(tot["dp_ref_bytes"])
        n_all = sum(x[2] for x in sizes)
        step_s = elapsed / steps
        S1, J1 = S / n_chunks, J / n_chunks          # per launch (= per chunk)
        ms_bytes = 128 * S1 + 64 * J1                # algorithmic bytes of the LF stage
        ms_s = kern_launch[0] / 1e3
        achieved = ms_bytes / ms_s / 1e9 if ms_s > 0 else 0.0
        layout_bytes = 73 * S1      # what the move-structure layout itself needs per LF step: one 64-byte fast row, one 8-byte pointer store, 1/8 of a packed pattern word
        path_bytes = 128 * S + 64 * J + 128 * P + C + R
        aligned_all = sum(x[0] for x in sizes)          # per step, all ranks
        value = aligned_all * steps / elapsed            # the metric counts ALIGNED reads; every read of the set goes through the path (reads_per_s)
        # the align stage's own bound: VALU issue.  A DP cell PAIR (two problems side by side in the 16-bit halves) costs 31 instructions of one wavefront, counted (profiles/r05k/summary.txt:
        # SQ_INSTS_VALU of the dp_lane kernels over the cell slots they stepped through; 28 of them in the row loop), i.e. 15.5 per cell; a wavefront
        # instruction occupies its SIMD for 4 cycles: peak = SIMDs x clock / 4 x 128 cells / 31.  The instruction count comes from the last committed counter pass (counters
        # cannot be read inside this run) and is labelled as static.
        simds, clk = 1024, 2.4e9
        dp_peak_tcups = simds * clk / 4 * 128 / 31 / 1e12
        dp_s = grp["dp_lane"]
        out = {
            "metric": "aligned reads/s (whole node), %d bp SE, mouse-chr19-scale x%d-haplotype index" % (L, args.haps),
            "value": value, "unit": "reads/s", "reads_per_s": n_all * steps / elapsed, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u64/int32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[%d]: mouse-chr19-scale index (%d bp base%s + %d haplotypes, %s, n=%d, r=%d), "
                                   "%s x %d bp reads resident in HBM -> MEM seeding + staged align kernels (chaining, lane-per-problem ksw2 DP, traceback, "
                                   "lift-over, SAM lines, lines gathered in read order) -> SAM text in pinned host memory, one transfer per sub-batch (%d host threads per GPU stand by for reads handed back)"
                                   % (3 if sharded else 2, args.base_len, " with %g interspersed repeats" % args.repeats if args.repeats else "", args.haps,
                                      "FASTA-built: null lifts" if args.fasta_index else "ref+VCF -H12 style: haplotypes lift onto the reference contig", fi_n, fi_r,
                                      ("one set of %d sharded over %d rank(s) by contiguous ranges, %d resident chunk(s) of <= %d per rank" % (total, world, n_chunks, args.reads)) if sharded else ("%d per GPU" % args.reads), L, threads),
                       "reads_per_gpu": n_mine, "total_reads": n_all, "chunks_per_rank": n_chunks, "contexts_in_flight": inflight, "read_len": L, "parallelism": "reads sharded x%d, index replicated" % world,
                       "launched_by": "bench.py" if os.environ.get("MONI_BENCH_SELF_LAUNCHED") else ("torch.distributed.run / external launcher" if world > 1 else "single process")},
            # headline: the bytes the layout itself has to move per launch (one 64-byte fast row + one 8-byte pointer store + 1/8 of a packed pattern
            # word per LF step = 73 S; the PMC passes count 1.33x that).  SURVEY.md 8(d)'s figure (128 S + 64 J: two requests per step, the reference's
            # data-structure model) is kept beside it, labelled - it credits the kernel with bytes the fast-row layout never moves (VERDICT r2, item 3 iii)
            "roofline": {"bound": "hbm", "kernel": "ms_lf_kernel", "achieved": layout_bytes / ms_s / 1e9 if ms_s > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": layout_bytes / ms_s / 1e9 / HBM_PEAK_GBS if ms_s > 0 else 0.0, "traffic": None,
                         "model": "layout: 73 bytes per LF step (64-byte fast row, threshold jumps included + 8-byte pointer store + 1 byte of packed pattern)",
                         "bytes_per_launch": layout_bytes, "avg_launch_ms": kern_launch[0], "per_read_bytes": layout_bytes / max(1, n_first),
                         "survey_8d": {"bytes_per_launch": ms_bytes, "formula": "128 S + 64 J (SURVEY.md 8(d): two 64-byte requests per LF step, one per threshold jump)",
                                       "achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "per_read_bytes": ms_bytes / max(1, n_first)},
                         "traffic_static_from": "profiles/r05k/pmc_hbm.csv (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of this "
                                                "workload: 26.64 GB fetched + 2.40 GB written = 29.04 GB per launch of 1 M reads = 1.33 x the layout's 21.9 GB, 0.62 x the survey's 47.18 GB; "
                                                "counters cannot be read inside this run, so `traffic` stays null)",
                         "note": "the path's HBM-bound kernel (LF / threshold-jump stage of seeding), one launch per resident chunk inside the timed region, HIP events on its own stream "
                                 "(with --inflight 2 another step's kernels run beside it: `kernel_alone` is the launch by itself); what bounds it is the rate at which random 64-byte lines "
                                 "come back to a CU's L1 (~107 reads in flight per CU at ~1000 cycles each: profiles/r04r/pmc_l2_latency.txt), not bytes"},
            "whole_path": {"bytes_per_read": path_bytes / max(1, n_mine), "formula": "128 S + 64 J + 128 P + C + R (SURVEY.md 8(d)), all counted by the kernels in this run",
                           "S_lf_steps": S, "J_threshold_jumps": J, "P_phi_steps": P, "C_text_bytes": C, "R_dp_target_bytes": R,
                           "GB/s": path_bytes / step_s / 1e9, "frac_of_hbm_peak": path_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                           "note": "rank 0's reads over rank 0's step time: the step as a whole is bound by dependent-access latency in the chaining and record kernels, not by HBM"},
            "align": {"kernels_ms_per_step_summed": {k: v * 1e3 for k, v in grp.items()}, "span_ms_per_step": stage["align_kernels_span"] * 1e3,
                      "note": "HIP-event times of the staged align kernels by group, summed over the sub-batches (two launch streams overlap, so the sum exceeds the span)",
                      "dp_problems": tot["dp_tasks"], "dp_cells": tot["dp_cells"],
                      "dp_gcups_in_kernel": tot["dp_cells"] / grp["dp_lane"] / 1e9 if grp["dp_lane"] > 0 else None,
                      "roofline": {"bound": "valu", "kernel": "dp_lane_kernel / dp_band_kernel / dp_wave_kernel (all instances)", "unit": "TCUPS",
                                   "cells": tot["dp_cells"], "cells_after_cut": tot["dp_cells_cut"], "cell_slots_run": tot["dp_slots"],
                                   "padding": {"useful_over_slots": tot["dp_cells_cut"] / tot["dp_slots"] if tot["dp_slots"] else None,
                                               "note": "cells: qlen x tlen of every ksw_extz2_sse call as the reference poses it; cells_after_cut: after the target rows of an extension that cannot hold its result are dropped "
                                                       "(af_build_cand); cell_slots_run: 128 problems x the chunk's longest query x the target rows of its passes - what the kernels actually stepped through"},
                                   "achieved": tot["dp_cells"] / dp_s / 1e12 if dp_s > 0 else None, "peak": dp_peak_tcups,
                                   "frac": tot["dp_cells"] / dp_s / 1e12 / dp_peak_tcups if dp_s > 0 else None,
                                   "peak_model": "%d SIMDs x %.1f GHz / 4 cycles per wavefront instruction x 128 cells per instruction (64 lanes x two 16-bit halves) / 31 instructions per cell pair" % (simds, clk / 1e9),
                                   "achieved_note": "reference cells over the summed HIP-event time of the DP kernels (launches of two streams overlap: a lower bound of the in-kernel rate)",
                                   "valu_static_from": "profiles/r05k/summary.txt (rocprofv3 --pmc SQ_INSTS_VALU of this workload with every problem on dp_lane_kernel: 4.22e9 wavefront instructions for 1.74e10 cell slots = 31.1 per cell pair)"},
                      "reads_taken_by_general_kernel": tot["kernel_fallback"], "reads_handed_to_host_pipeline": tot["handed_back"],
                      "handed_over_because": {k: v // steps for k, v in why.items()}},
            "stages_s_per_step": stage,
            "n_rate": args.n_rate,
            "aligned_per_step": tot["aligned"], "sam_bytes_per_step": sam_len,
            "aligned_all_ranks": aligned_all,
            "setup_s": {"total": setup_s, "phases": [[n_, round(d_, 2)] for n_, d_ in phases], "note": "rank 0's wall time in front of the timed region"},
            "kernels_ms": {"ms_lf": kern_launch[0], "mem_count": kern_launch[1], "mem_emit": kern_launch[2], "occ_count": kern_launch[3], "occ_fill": kern_launch[4],
                           "seeding_whole": kern_launch[6], "note": "per launch (one launch per resident chunk)"},
            "seeding": {"workload": "BASELINE.json configs[1]: MEM seeding stage alone on rank 0's first resident chunk (%d reads)" % n_first,
                        "value": n_first / seed_s, "unit": "reads/s", "ms_per_pass": seed_s * 1e3,
                        "work_per_pass": {"lf_steps": int(S1), "threshold_jumps": int(J1), "phi_steps": P // n_chunks, "text_bytes": C // n_chunks, "mems": n_mems, "occs": n_occs}},
        }
        if gather:
            out["gather"] = gather
            out["value_with_gather"] = n_all / (step_s + gather["seconds"])
        if from_host:
            out["from_host"] = from_host
        if scaling_base:
            out["scaling_base"] = scaling_base
        if single:
            out["single_context"] = single
            if single["ms_lf_kernel_ms"] > 0:          # the headline leg's launches share the GPU with another step's kernels: the kernel by itself
                a_s = single["ms_lf_kernel_ms"] / 1e3
                out["roofline"]["kernel_alone"] = {"avg_launch_ms": single["ms_lf_kernel_ms"], "achieved": layout_bytes / a_s / 1e9, "frac": layout_bytes / a_s / 1e9 / HBM_PEAK_GBS,
                                                   "note": "last launch of the single-context leg (HIP events): no other step's kernels beside it"}
        out["host"] = {"cpus_usable": host_cpus(), "cpu_count": os.cpu_count(), "host_threads_per_gpu": threads}
        out["device_memory_GB"] = {"used_after_timed_region": (mem_total - mem_free) / 1e9, "total": mem_total / 1e9,
                                   "note": "rank 0's GPU: index image + the working memory of its %d context(s)" % inflight}
        if world == 1 and not args.no_cpu:
            from oracle import orc as _orc          # the CPU baseline / at-scale checker: the only use of oracle/ in this file
            oidx = _orc.OracleIndex(fi=fi)          # (rank 0 holds the flat index: built here or mapped from the cache file)
            cpu_threads = host_cpus()
            nm, no, ql, nb = chunk[0]
            rd, ob = reads[:nb], offs[:nb + 1]
            # full path on the CPU (oracle/align.hpp), bounded sample; the same sample is an at-scale SAM identity check
            probe = min(2000, nb)
            t1 = time.perf_counter()
            _orc.align_batch(oidx, rd[:probe].reshape(-1), ob[:probe + 1], nm[:int(no[probe])], no[:probe + 1], ql[:probe * L], threads=cpu_threads)
            rate = probe / (time.perf_counter() - t1)
            n_cpu = int(max(probe, min(nb, rate * args.cpu_seconds)))
            t1 = time.perf_counter()
            wsam, wc = _orc.align_batch(oidx, rd[:n_cpu].reshape(-1), ob[:n_cpu + 1], nm[:int(no[n_cpu])], no[:n_cpu + 1], ql[:n_cpu * L], threads=cpu_threads)
            dtc = time.perf_counter() - t1
            gsam, _ = ctx.align_batch(rd[:n_cpu].reshape(-1), ob[:n_cpu + 1], nm[:int(no[n_cpu])], no[:n_cpu + 1], ql[:n_cpu * L], host_threads=threads)
            out["cpu_baseline"] = {"value": n_cpu / dtc, "unit": "reads/s", "cores": cpu_threads, "kind": "port",
                                   "sample": "first %d reads of the same batch, whole SE path, oracle/align.hpp with %d threads" % (n_cpu, cpu_threads),
                                   "sam_identical_on_sample": bool(gsam == wsam)}
            # BASELINE.json configs[0]: the CPU path on one thread (the plumbing / SAM-diff baseline), small sample
            n1 = min(2000, n_cpu)
            t1 = time.perf_counter()
            w1, _ = _orc.align_batch(oidx, rd[:n1].reshape(-1), ob[:n1 + 1], nm[:int(no[n1])], no[:n1 + 1], ql[:n1 * L], threads=1)
            out["cpu_baseline"]["single_thread"] = {"value": n1 / (time.perf_counter() - t1), "unit": "reads/s", "cores": 1,
                                                    "sample": "first %d reads" % n1, "same_text_as_all_threads": bool(w1 == wsam[:len(w1)])}
            # seeding stage alone on the CPU, same bounded way
            t1 = time.perf_counter()
            oidx.seed_batch(rd[:probe].reshape(-1), ob[:probe + 1], 25, True, 1000, threads=cpu_threads)
            rate = probe / (time.perf_counter() - t1)
            n_cs = int(max(probe, min(nb, rate * args.cpu_seconds * 0.5)))
            t1 = time.perf_counter()
            want = oidx.seed_batch(rd[:n_cs].reshape(-1), ob[:n_cs + 1], 25, True, 1000, threads=cpu_threads)
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
    return 0


# ---- the paired-end path -----------------------------------------------------------------------------------
def run_paired(args, rank, world, dist, coll_dev, backend, pg, fi, idx, ctx, phases, t_start) -> int:
    """One "step" = moni_pe_align_batch over this rank's pairs (host memory -> two SAM records per pair in host memory; the paired entry points
    have no resident form), fragment model learnt once on rank 0's first batches of 512 pairs (the reference's single learner,
    align_reads_dispatcher.hpp:356-389) and broadcast."""
    import torch
    from moni_align_amd import capi, dist as mdist, synth
    L = args.read_len
    n_all = args.pairs * world
    mates, ins = synth.make_pairs(pg, n_all, L, seed=350)
    names, noff = synth.make_pair_names(n_all)
    lo, hi = mdist.shard_range(n_all, rank, world)
    model = capi.PeModelC()
    zsec = 1 if args.secondary_chains else 0
    vals = torch.zeros(8, dtype=torch.float64)
    if rank == 0:
        at = 0
        while not model.complete and at < n_all:
            e = min(n_all, at + 512)
            ctx.pe_learn(mates[2 * at:2 * e].reshape(-1), np.arange(0, (2 * (e - at) + 1) * L, L, dtype=np.uint64), model, secondary_chains=zsec)
            at = e
        vals = torch.tensor([model.mean, model.std_dev, model.variance, model.sample_variance, model.m2, float(model.count), float(model.complete), 0.0], dtype=torch.float64)
    if dist is not None:          # learn once, broadcast (every rank aligns with the same model)
        v = vals.to(coll_dev)
        dist.broadcast(v, src=0)
        vals = v.cpu()
        model.mean, model.std_dev, model.variance, model.sample_variance, model.m2 = (float(x) for x in vals[:5])
        model.count, model.complete = int(vals[5]), int(vals[6])
    seq = mates[2 * lo:2 * hi].reshape(-1)
    offs = np.arange(0, (2 * (hi - lo) + 1) * L, L, dtype=np.uint64)
    nm = names[int(noff[2 * lo]):int(noff[2 * hi])]
    no = (noff[2 * lo:2 * hi + 1] - noff[2 * lo]).astype(np.uint64)
    ql = np.full(seq.size, ord("I"), np.uint8)
    threads = max(1, host_cpus() // max(1, world))

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    st = None
    # reads handed over in host memory (moni_pe_align_stream: the upload inside the call): reported beside the headline, never as `value`
    from_host = None
    if not args.no_from_host:
        ctx.pe_align(seq, offs, nm, no, ql, model, host_threads=threads, want_text=False, stream=True, secondary_chains=zsec)
        t1 = time.perf_counter()
        for _ in range(max(1, args.steps)):
            ctx.pe_align(seq, offs, nm, no, ql, model, host_threads=threads, want_text=False, stream=True, secondary_chains=zsec)
        dt = (time.perf_counter() - t1) / max(1, args.steps)
        from_host = {"value": (hi - lo) / dt, "unit": "pairs/s (this rank)", "ms_per_batch": dt * 1e3, "note": "moni_pe_align_stream: mates, names, qualities in pageable host memory, upload inside the timed call"}
        if world == 1:          # a streaming caller keeps several contexts per GPU going (moni-hip-align -1/-2 runs three): one's upload, seeding and tail beside the other's paired kernels
            ctx_b = capi.Ctx(idx)
            ctx_b.pe_align(seq, offs, nm, no, ql, model, host_threads=threads, want_text=False, stream=True, secondary_chains=zsec)
            reps = max(2, min(6, args.steps))

            def pe_worker(cx):
                for _ in range(reps):
                    cx.pe_align(seq, offs, nm, no, ql, model, host_threads=max(1, threads // 2), want_text=False, stream=True, secondary_chains=zsec)
            th = [threading.Thread(target=pe_worker, args=(cx,)) for cx in (ctx, ctx_b)]
            t1 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            two_s = (time.perf_counter() - t1) / (2 * reps)
            ctx_b.close()
            from_host["two_contexts"] = {"value": (hi - lo) / two_s, "unit": "pairs/s", "ms_per_batch": two_s * 1e3, "note": "two contexts on the GPU, one caller thread each"}
    # the headline: the mates resident in HBM when the timed region starts (moni_reads_upload), names and qualities host buffers as in moni_align_run
    inflight = max(1, args.inflight)
    ctxs = [ctx] + [capi.Ctx(idx) for _ in range(inflight - 1)]          # --inflight: the same mates resident in every context, one caller thread each
    th_step = max(1, threads // inflight)
    for cx in ctxs:
        cx.upload(seq, offs)
        for _ in range(args.warmup):
            cx.pe_align_run(nm, no, ql, model, host_threads=th_step, want_text=False, secondary_chains=zsec)
    sync_all()
    left, last, lk = [args.steps], {}, threading.Lock()

    def stepper(cx):          # exactly args.steps passes in all: a context takes the next one as soon as it is through with its last
        while True:
            with lk:
                if left[0] <= 0:
                    return
                left[0] -= 1
            r_ = cx.pe_align_run(nm, no, ql, model, host_threads=th_step, want_text=False, secondary_chains=zsec)      # the text is in the context's pinned host buffer; no copy into a Python object
            with lk:
                last["r"] = r_
    t0 = time.perf_counter()
    if inflight == 1:
        stepper(ctx)
    else:
        th = [threading.Thread(target=stepper, args=(cx,)) for cx in ctxs]
        for t in th:
            t.start()
        for t in th:
            t.join()
    sync_all()
    elapsed = mdist.max_over_ranks(time.perf_counter() - t0, dist, coll_dev)
    sam_len, st = last["r"]
    single = None
    if inflight > 1 and not args.no_single_context:
        n1s = max(1, min(3, args.steps))
        t1 = time.perf_counter()
        for _ in range(n1s):
            _, st = ctx.pe_align_run(nm, no, ql, model, host_threads=threads, want_text=False, secondary_chains=zsec)
        sync_all()
        e1 = mdist.max_over_ranks(time.perf_counter() - t1, dist, coll_dev)
        single = {"ms_per_step": e1 / n1s * 1e3, "steps": n1s, "pairs_per_s_this_rank": (hi - lo) * n1s / e1, "note": "the same resident mates, one context by itself, after the timed region; the stage times and the roofline's launch time below are this leg's"}
    for cx in ctxs[1:]:
        cx.close()
    lf_ms = ctx.kernel_ms(0)                      # ms_lf_kernel of the last step (one launch over the 2 N mates), HIP events on its own stream
    S_lf, J_lf = (int(x) for x in ctx.counters()[:2])
    sizes = mdist.gather_counts([st["aligned"], hi - lo], dist, coll_dev)
    if rank == 0:
        steps = max(1, args.steps)
        step_s = elapsed / steps
        out = {"metric": "aligned read pairs/s (whole node), 2 x %d bp PE, mouse-chr19-scale x%d-haplotype index, orphan recovery on%s" % (L, args.haps, ", -Z (secondary chains)" if zsec else ""),
               "value": n_all / step_s, "unit": "pairs/s", "reads_per_s": 2 * n_all / step_s, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_s * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64/int32", "data": "synthetic",
               "config": {"workload": "SURVEY.md 8(f)-2: paired-end path on the BASELINE.json configs[2] index (%d bp base + %d haplotypes, n=%d, r=%d): %d FR pairs of 2 x %d bp per GPU "
                                      "(insert 350 +- 30, 0.5 %% substitutions) resident in HBM -> seeding kernels over the 2 N mates + staged paired kernels (wave per pair plan, lane per DP problem, select, finish, both SAM lines written and "
                                      "ordered on the GPU; pe_align_kernel for the pairs they hand over: orphan recovery) -> two SAM records per pair in pinned host memory" % (args.base_len, args.haps, idx.n, idx.r, args.pairs, L),
                          "pairs_per_gpu": hi - lo, "read_len": L, "contexts_in_flight": inflight, "parallelism": "pairs sharded x%d, index replicated, fragment model learnt on rank 0 and broadcast" % world},
               "model": {"count": int(model.count), "mean": model.mean, "std_dev": model.std_dev, "complete": bool(model.complete)},
               "aligned_pairs_all_ranks": sum(x[0] for x in sizes), "stages_s_per_step": {"seed": st["t_seed"], "kernel_and_copies": st["t_dp"], "host_finish": st["t_host"]},
               "dp_problems": st["dp_tasks"], "dp_cells": st["dp_cells"], "pairs_through_host_pipeline": st["handed_back"],
               "pairs_taken_by_pe_align_kernel": st["kernel_fallback"], "handed_over_because": st["handover_why"],
               # the path's HBM-bound kernel is the single-end path's: the LF / threshold-jump stage of seeding over the 2 N mates (same model as the default line)
               "roofline": {"bound": "hbm", "kernel": "ms_lf_kernel", "achieved": 73 * S_lf / (lf_ms / 1e3) / 1e9 if lf_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": 73 * S_lf / (lf_ms / 1e3) / 1e9 / HBM_PEAK_GBS if lf_ms > 0 else 0.0, "traffic": None, "model": "layout: 73 bytes per LF step", "avg_launch_ms": lf_ms,
                            "survey_8d": {"bytes_per_launch": 128 * S_lf + 64 * J_lf, "frac": (128 * S_lf + 64 * J_lf) / (lf_ms / 1e3) / 1e9 / HBM_PEAK_GBS if lf_ms > 0 else 0.0}},
               "pairs_handed_over": st["kernel_fallback"],          # by the staged kernels to pe_orphan_kernel (orphan recovery: counted as loop_depends_on_score; capacities) "handed_over_because": {k: v for k, v in st.get("handover_why", {}).items() if v},
               "host": {"cpus_usable": host_cpus(), "host_threads_per_gpu": threads}}
        if from_host is not None:
            out["from_host"] = from_host
        if single is not None:
            out["single_context"] = single
        if world == 1 and not args.no_cpu:
            from oracle import orc as _orc          # the CPU baseline / at-scale checker
            oidx = _orc.OracleIndex(fi=fi)
            cpu_threads = host_cpus()

            def cpu_pe(a, b):          # the oracle's paired path over pairs [a, b), st_align's order (learns on its own first batches of 512)
                m1, m2 = mates[2 * a:2 * b:2], mates[2 * a + 1:2 * b:2]
                o1 = np.arange(0, (b - a + 1) * L, L, dtype=np.uint64)
                n1 = b"".join(bytes(names[int(noff[2 * p]):int(noff[2 * p + 1])]) for p in range(a, b)); n2 = b"".join(bytes(names[int(noff[2 * p + 1]):int(noff[2 * p + 2])]) for p in range(a, b))
                no1 = np.zeros(b - a + 1, np.uint64); no1[1:] = np.cumsum([int(noff[2 * p + 1] - noff[2 * p]) for p in range(a, b)])
                no2 = np.zeros(b - a + 1, np.uint64); no2[1:] = np.cumsum([int(noff[2 * p + 2] - noff[2 * p + 1]) for p in range(a, b)])
                q = np.full((b - a) * L, ord("I"), np.uint8)
                return _orc.align_pe(oidx, np.ascontiguousarray(m1).reshape(-1), o1, np.ascontiguousarray(m2).reshape(-1), o1, np.frombuffer(n1, np.uint8), no1,
                                     np.frombuffer(n2, np.uint8), no2, q, q, b_size=512, find_orphan=True, secondary_chains=bool(zsec))
            n0 = min(3000, hi - lo)
            t1 = time.perf_counter()
            want, ost = cpu_pe(0, n0)
            rate1 = n0 / (time.perf_counter() - t1)
            nseq0 = int(offs[2 * n0] - offs[0])
            got, _ = ctx.pe_align(seq[:nseq0], offs[:2 * n0 + 1], nm[:int(no[2 * n0])], no[:2 * n0 + 1], None if ql is None else ql[:nseq0], model, host_threads=threads, secondary_chains=zsec)
            per = int(max(1, min((hi - lo) // cpu_threads, max(2000, rate1 * args.cpu_seconds))))          # never past this rank's pairs (few pairs on many cores: shorter slices)
            res = [None] * cpu_threads
            th = [threading.Thread(target=lambda k=k: res.__setitem__(k, cpu_pe(k * per, (k + 1) * per))) for k in range(cpu_threads)]
            t1 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            dtc = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": per * cpu_threads / dtc, "unit": "pairs/s", "cores": cpu_threads, "kind": "port",
                                   "sample": "%d slices of %d pairs, oracle/align_pe.hpp with orphan recovery, one thread per slice (each learns its own model as a run over that slice would)" % (cpu_threads, per),
                                   "single_thread": {"value": rate1, "unit": "pairs/s", "cores": 1, "sample": "first %d pairs" % n0},
                                   "sam_identical_on_sample": bool(got == want), "model_identical": bool(ost["ins_mean"] == model.mean and ost["ins_std_dev"] == model.std_dev)}
        print(json.dumps(out), flush=True)
    ctx.close()
    idx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))            # the parent: starts the ranks, never initialises HIP
    sys.exit(dry_run(args) if args.dry_run else run_rank(args))


if __name__ == "__main__":
    main()
