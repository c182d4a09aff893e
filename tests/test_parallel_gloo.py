"""N > 1 path on CPU: two gloo processes shard the directed pairs by source image, each links its shard
(the oracle stands in for the device here — this test is about the sharding / gather logic, which is
plain host code), and the gathered result equals the single-process result pair for pair."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from opencalibration_amd import parallel, synth  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _link_records(grid, pairs):
    from oracle import pyoracle

    subsets = {}
    for i in parallel.images_needed(pairs):
        loc, st, _, _ = grid.image(i)
        subsets[i] = pyoracle.subsample(loc, st, 40.0, int(grid.num_sparse[i]))
    out = []
    for a, b in pairs:
        la, _, da, _ = grid.image(a)
        lb, _, db, _ = grid.image(b)
        r = pyoracle.link_pair(la, da, subsets[a], lb, db, subsets[b], grid.model, grid.model)
        out.append((a, b, (r["inliers"].tobytes(), r["H"].tobytes(), r["score"])))
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grid = synth.make_grid(2, 3, feats=300, seed=31)
    pairs = parallel.knn_pairs(grid.position[:, :2])
    mine = parallel.shard_pairs(pairs, grid.n_images, rank, world)
    merged = parallel.gather_edges(_link_records(grid, mine))
    if rank == 0:
        q.put((len(mine), merged))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_link_equals_single_process():
    grid = synth.make_grid(2, 3, feats=300, seed=31)
    pairs = parallel.knn_pairs(grid.position[:, :2])
    expected = sorted(_link_records(grid, pairs), key=lambda r: (r[0], r[1]))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    n_mine, merged = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert 0 < n_mine < len(pairs)
    assert merged == expected


def test_blocks_partition_the_sources():
    for n, w in ((1000, 8), (10, 3), (7, 8), (1, 2)):
        seen = []
        for r in range(w):
            lo, hi = parallel.source_block(n, r, w)
            seen += list(range(lo, hi))
        assert seen == list(range(n))
    grid = synth.make_grid(3, 4, feats=64, seed=2)
    pairs = parallel.knn_pairs(grid.position[:, :2])
    parts = [parallel.shard_pairs(pairs, grid.n_images, r, 4) for r in range(4)]
    assert sorted(p for part in parts for p in part) == sorted(pairs)
    assert sum(len(p) for p in parts) == len(pairs)


def _gather_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    stats = {}
    mine = (np.arange(1000 * (rank + 1) + 3 * rank) % 251).astype(np.uint8)   # ragged sizes, not multiples of 8
    parts = parallel.all_gather_bytes(mine, stats=stats)
    empty = parallel.all_gather_bytes(np.zeros(0, np.uint8))
    if rank == 1:
        q.put(([p.tobytes() for p in parts], [len(e) for e in empty], stats, [p.ctypes.data % 8 for p in parts]))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_bytes_ragged():
    """The byte transport of the sharded survey (parallel.all_gather_bytes): ragged buffers, 8-byte aligned slices."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    parts, empty, stats, align = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for r in range(3):
        assert parts[r] == (np.arange(1000 * (r + 1) + 3 * r) % 251).astype(np.uint8).tobytes()
    assert empty == [0, 0, 0] and align == [0, 0, 0]
    assert stats == {"exchanges": 1, "bytes_gathered": sum(1000 * (r + 1) + 3 * r for r in range(3))}
