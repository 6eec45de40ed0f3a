"""world_size-2 gloo test of the N>1 plumbing (shard bounds, in-order gather, max-over-ranks, barrier)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from defectdetection_viaobjectdetection_amd.sharding import gather_in_order, max_over_ranks, shard_bounds


def test_shard_bounds_cover_and_order():
    for n in (0, 1, 7, 32, 33):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_items):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b = shard_bounds(n_items, world, rank)
        local = [("img%03d" % i, i * i) for i in range(a, b)]       # stand-in for per-image results
        dist.barrier()
        allr = gather_in_order(local)
        assert [x[0] for x in allr] == ["img%03d" % i for i in range(n_items)]
        assert [x[1] for x in allr] == [i * i for i in range(n_items)]
        t = max_over_ranks(1.0 + rank)
        assert t == float(world)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_roundtrip():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, 7), nprocs=2, join=True)
