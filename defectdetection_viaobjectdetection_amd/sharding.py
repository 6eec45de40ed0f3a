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


class GradBucketReducer:
    """Training-side exchange step: SUM all-reduce of the flat fp32 gradient buffer across ranks,
    issued in buckets while the backward pass is still producing the rest.

    The flat buffer is laid out in forward parameter order and the backward pass finishes parameters roughly from
    the end of the buffer towards its start, so the reducer tracks the contiguous *suffix* of finished elements and
    fires an asynchronous all-reduce (RCCL on its own stream; gloo in the CPU tests) each time the suffix has grown
    by at least ``bucket_bytes``.  xGMI rings are per-link bound, so few large buckets (default 32 MiB, i.e. 4 for
    s-seg, 4 for m-seg's 109 MB) beat many small ones.  Upstream's convention is kept: the summed loss is already
    scaled by the local batch size, ranks' gradients are SUMMED (DDP's mean x the trainer's world-size factor).
    """

    def __init__(self, flat, spans, bucket_bytes: int = 32 << 20, group=None):
        """flat: 1-D gradient tensor; spans: {name: (offset, numel)} covering ``flat`` exactly."""
        self.flat, self.group = flat, group
        self.spans = dict(spans)
        covered = sorted(self.spans.values())
        pos = 0
        for off, n in covered:
            if off != pos:
                raise ValueError("spans must tile the flat buffer without gaps or overlaps")
            pos += n
        if pos != flat.numel():
            raise ValueError("spans do not cover the flat buffer")
        self.bucket_elems = max(1, bucket_bytes // flat.element_size())
        self.reset()

    def reset(self) -> None:
        self._done = set()
        self._by_end = {off + n: (name, off) for name, (off, n) in self.spans.items()}
        self._suffix = self.flat.numel()      # elements [suffix, end) are finished
        self._sent = self.flat.numel()        # elements [sent, end) have been handed to all_reduce
        self._works = []
        self.launched = []                    # (lo, hi) of every bucket, for tests / logging

    def _fire(self, lo: int, hi: int) -> None:
        import torch.distributed as dist
        if hi <= lo:
            return
        self.launched.append((lo, hi))
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def mark_ready(self, name: str) -> None:
        """Call when the gradient of ``name`` is final for this step."""
        if name not in self.spans:
            raise KeyError(name)
        self._done.add(name)
        while self._suffix in self._by_end and self._by_end[self._suffix][0] in self._done:
            self._suffix = self._by_end[self._suffix][1]
        if self._sent - self._suffix >= self.bucket_elems:
            self._fire(self._suffix, self._sent)
            self._sent = self._suffix

    def finish(self) -> None:
        """Reduce whatever is left (everything, if mark_ready was never called) and wait for all buckets."""
        missing = [k for k in self.spans if k not in self._done]
        if missing and len(missing) != len(self.spans):
            raise RuntimeError(f"{len(missing)} gradients were never marked ready, e.g. {missing[:3]}")
        self._fire(0, self._sent)
        self._sent = 0
        for w in self._works:
            w.wait()
        self._works = []
