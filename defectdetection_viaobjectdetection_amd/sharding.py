"""Multi-GPU plumbing of the inference path: one process per GPU, the image batch is sharded across ranks
with NO data-path collective.  ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests) is used only for rendezvous, barriers, the max-over-ranks of a timing
and gathering per-image results back in input order."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_bounds(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, end) slice of ``n_items`` for ``rank``; the first ``n % world`` ranks get one extra."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad world_size / rank")
    q, r = divmod(n_items, world_size)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def gather_in_order(local_items: Sequence, group=None) -> List:
    """All ranks receive the concatenation of every rank's items in rank order (= input order for
    ``shard_bounds`` slices).  Falls back to a plain list when torch.distributed is not initialised."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return list(local_items)
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, list(local_items), group=group)
    return [x for p in parts for x in p]


def max_over_ranks(value: float, device=None, group=None) -> float:
    """MAX all-reduce of a scalar (the bench's elapsed time)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
