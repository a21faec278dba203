"""Multi-GPU plumbing: one process per GPU, reads sharded by contiguous ranges, index replicated.
torch.distributed carries the barrier, the MAX reduce of the elapsed time, the all-gather of per-rank result sizes and the
final gather of the per-rank SAM blocks to rank 0 in rank (= input) order, the order the reference's mt_align concatenates
its per-thread files in (include/aligner/align_reads_dispatcher.hpp:267-274).  Backend nccl = RCCL over xGMI on the GPU box,
gloo in the CPU tests."""
from __future__ import annotations

import os
from typing import List, Tuple

import torch


def env_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str, rank: int, world: int, local_rank: int = 0):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range [k*N/G, (k+1)*N/G) of reads for this rank (strong-scaling split of one input)."""
    return n_items * rank // world, n_items * (rank + 1) // world


def max_over_ranks(value: float, dist=None, device: str = "cpu") -> float:
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(counts: List[int], dist=None, device: str = "cpu") -> List[List[int]]:
    """all_gather of a small per-rank vector of result sizes (records, occurrences, ...)."""
    if dist is None:
        return [list(counts)]
    t = torch.tensor(counts, dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[int(x) for x in o.tolist()] for o in out]


def gather_sam(block, dist=None, device: str = "cpu"):
    """Gather every rank's SAM block (bytes or a uint8 tensor) on rank 0, concatenated in rank order.

    all_gather of the block sizes, then point-to-point: ranks > 0 send their block, rank 0 receives them into one buffer at
    their final offsets (7 peers -> 7 xGMI links in parallel on the GPU box; no ring).  Returns (tensor, sizes)
    on rank 0 and (None, sizes) elsewhere.  With device="cuda" the blocks travel GPU to GPU (RCCL needs device tensors)."""
    t = block if isinstance(block, torch.Tensor) else torch.frombuffer(bytearray(block), dtype=torch.uint8)
    if dist is None:
        return t, [int(t.numel())]
    t = t.to(device)
    rank, world = dist.get_rank(), dist.get_world_size()
    sizes = [row[0] for row in gather_counts([int(t.numel())], dist, device)]
    if rank != 0:
        if sizes[rank]:
            dist.send(t, dst=0)
        return None, sizes
    out = torch.empty(sum(sizes), dtype=torch.uint8, device=device)
    out[: sizes[0]] = t
    at = sizes[0]
    reqs = []
    for src in range(1, world):
        if sizes[src]:
            reqs.append(dist.irecv(out[at:at + sizes[src]], src=src))
        at += sizes[src]
    for r in reqs:
        r.wait()
    return out, sizes
