"""world_size-2 gloo run of the multi-GPU plumbing bench.py uses (sharding, barrier, MAX reduce, size gather),
plus the property that sharded seeding equals unsharded seeding (checked with the oracle on CPU)."""
import os

import numpy as np
import torch
import torch.multiprocessing as mp


def _worker(rank, world, port, n_reads, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from moni_align_amd import dist as md
    d = md.init("gloo", rank, world)
    lo, hi = md.shard_range(n_reads, rank, world)
    d.barrier()
    t = md.max_over_ranks(1.0 + rank, d)
    sizes = md.gather_counts([hi - lo, rank], d)
    q.put((rank, lo, hi, t, sizes))
    d.barrier()
    d.destroy_process_group()


def test_two_rank_plumbing():
    world, n_reads = 2, 1001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_reads, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n_reads          # contiguous, complete, disjoint
    assert all(abs(r[3] - 2.0) < 1e-12 for r in res)                                     # MAX over ranks
    assert all(r[4] == [[res[0][2] - res[0][1], 0], [res[1][2] - res[1][1], 1]] for r in res)


def test_sharded_equals_unsharded(small_case):
    from moni_align_amd import dist as md
    from oracle import orc
    o = orc.OracleIndex(small_case.path)
    L, N = 100, 64
    reads = small_case.synth.make_reads(small_case.pg, N, L, seed=31)
    offs = np.arange(0, (N + 1) * L, L, dtype=np.uint64)
    whole = o.seed_batch(reads.reshape(-1), offs, 25, True, 1000)
    parts = []
    for rank in range(3):
        lo, hi = md.shard_range(N, rank, 3)
        parts.append(o.seed_batch(reads[lo:hi].reshape(-1), offs[: hi - lo + 1], 25, True, 1000))
    assert np.array_equal(np.concatenate([p["pos"] for p in parts]), whole["pos"])
    assert np.array_equal(np.concatenate([p["occs"] for p in parts]), whole["occs"])
    assert sum(len(p["len"]) for p in parts) == len(whole["len"])


def _gather_worker(rank, world, port, path, n_reads, L, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from moni_align_amd import dist as md, synth, index_build
    from oracle import orc
    d = md.init("gloo", rank, world)
    fi = index_build.FlatIndex.load(path)
    o = orc.OracleIndex(fi=fi)
    pg = synth.make_pangenome(4000, 4, site_spacing=250)
    reads = synth.make_reads(pg, n_reads, L, seed=77)
    names, noff = orc.make_names(n_reads)
    lo, hi = md.shard_range(n_reads, rank, world)          # one read set, sharded by contiguous ranges: BASELINE.json configs[3]'s arrangement
    offs = np.arange(0, (hi - lo + 1) * L, L, dtype=np.uint64)
    nm = names[int(noff[lo]):int(noff[hi])]
    no = (noff[lo:hi + 1] - noff[lo]).astype(np.uint64)
    block, _ = orc.align_batch(o, reads[lo:hi].reshape(-1), offs, nm, no, None)
    got, sizes = md.gather_sam(block, d)
    q.put((rank, bytes(got.numpy().tobytes()) if got is not None else None, sizes))
    d.barrier()
    d.destroy_process_group()


def test_sam_gather_equals_unsharded(small_case):
    """The gather of the per-rank SAM blocks on rank 0 (sizes by all-gather, blocks by send/recv) is byte for byte the text of
    the unsharded run: the SE path has no cross-read state (aligner_ksw2.hpp:314-325), ranks hold contiguous read ranges."""
    from oracle import orc
    world, n_reads, L = 2, 41, 100
    o = orc.OracleIndex(small_case.path)
    reads = small_case.synth.make_reads(small_case.pg, n_reads, L, seed=77)
    names, noff = orc.make_names(n_reads)
    whole, _ = orc.align_batch(o, reads.reshape(-1), np.arange(0, (n_reads + 1) * L, L, dtype=np.uint64), names, noff, None)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, small_case.path, n_reads, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r[0], r) for r in (q.get(timeout=180) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1][1] is None and res[0][1] == whole
    assert sum(res[0][2]) == len(whole) and res[0][2] == res[1][2]
