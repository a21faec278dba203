"""Multi-GPU plumbing: one process per GPU, reads sharded by contiguous ranges, index replicated, no data-path
collective.  torch.distributed is used only for the barrier, the MAX reduce of the elapsed time and the
final gather of per-rank result sizes (backend nccl = RCCL on the GPU box, gloo in the CPU tests)."""
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
