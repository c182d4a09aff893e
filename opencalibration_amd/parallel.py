"""Multi-GPU sharding of the link stage: directed image pairs are independent units, so
ranks take contiguous blocks of *source images* (grid order keeps >= 90 % of a block's neighbours
local), every rank holds all descriptors it needs, and no data-path collective runs; rank 0 gathers
the per-pair results and LinkStage::finalize's deterministic sort (link_stage.cpp:123-127) restores
the serial order.  One process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in
the CPU tests)."""
import numpy as np


def source_block(n_images, rank, world):
    """Contiguous block of source-image indices owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_images, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def knn_pairs(position_xy, k=10):
    """Directed pairs (a, b): b among the k nearest of a including a itself, a != b
    (LinkStage::init, src/pipeline/link_stage.cpp:22-38)."""
    xy = np.asarray(position_xy, np.float64)
    d = ((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1)
    knn = np.argsort(d, axis=1, kind="stable")[:, :k]
    return [(a, int(b)) for a in range(len(xy)) for b in knn[a] if b != a]


def shard_pairs(pairs, n_images, rank, world):
    """Pairs whose source image lies in this rank's block, in the global (serial) order."""
    lo, hi = source_block(n_images, rank, world)
    return [p for p in pairs if lo <= p[0] < hi]


def images_needed(pairs):
    """Images a rank must hold descriptors for: its sources plus the halo of their neighbours."""
    return sorted({a for a, _ in pairs} | {b for _, b in pairs})


def gather_edges(local_records, group=None):
    """All ranks' per-pair records on every rank, merged into the serial order of
    LinkStage::finalize: sort by (source, dest).  `local_records`: list of (source, dest, payload)."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        merged = list(local_records)
    else:
        out = [None] * dist.get_world_size(group)
        dist.all_gather_object(out, list(local_records), group=group)
        merged = [r for part in out for r in part]
    merged.sort(key=lambda r: (r[0], r[1]))
    return merged


class _DeviceBytes:
    """Zero-copy view of `nbytes` of device memory at `ptr` for torch (CUDA array interface, version 2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def relax_exchange(group=None):
    """The exchange step of a sharded single-group relax (ochip_relax_set_shard, include/ochip.h) on
    torch.distributed: every rank's slice of the per-pair record arrays is all-gathered in place.  Backend
    "nccl" (RCCL over xGMI) gathers device to device; "gloo" stages the slices through host memory (tests).
    Returns the ctypes callback to pass as Graph.relax_ground_plane(..., shard=(rank, world, callback)); keep a
    reference to it for as long as the relax runs."""
    import torch
    import torch.distributed as dist

    from .host import RELAX_EXCHANGE_FN

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    on_device = dist.get_backend(group) == "nccl"

    def gather(ptr, per_rank):
        if not ptr or per_rank == 0:
            return
        full = torch.as_tensor(_DeviceBytes(ptr, per_rank * world), device="cuda")
        mine = full[rank * per_rank:(rank + 1) * per_rank].clone()
        if on_device:
            dist.all_gather_into_tensor(full, mine, group=group)
        else:
            host = mine.cpu()
            parts = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(parts, host, group=group)
            full.copy_(torch.cat(parts))

    def callback(_user, acc, acc_bytes, cost, cost_bytes, fail, fail_bytes):
        try:
            gather(acc, acc_bytes)
            gather(cost, cost_bytes)
            gather(fail, fail_bytes)
            torch.cuda.synchronize()
            return 0
        except Exception as ex:  # a Python exception must not unwind through the C caller
            import sys
            print(f"relax exchange failed on rank {rank}: {ex!r}", file=sys.stderr, flush=True)
            return 1

    return RELAX_EXCHANGE_FN(callback)


def relax_exchange_rccl(ctx, group=None):
    """The native transport of the sharded relax: an RCCL communicator owned by libochip on `ctx` (the context the relax
    runs on), its all-gathers enqueued on that context's stream by the library itself - no Python in the solve loop, no
    host wait.  torch.distributed only carries the 128-byte ncclUniqueId from rank 0 to the others (any backend).
    Returns the capi.RcclComm to pass as Graph.relax_ground_plane(..., shard=(rank, world, comm)); close() it afterwards."""
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    box = [ctx.rccl_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    return ctx.rccl_comm(box[0], rank, world)


def link_sharded(make_graph, ctx, group=None):
    """One survey's link stage over the ranks of `group` on the DEVICE path.  `make_graph()` builds the survey's graph with
    its nodes (all images' features: the descriptor sets are small) and no edges.  Every rank links
    the directed pairs of its contiguous block of source images with LinkStage on its own GPU
    (host.Graph.link(node_ids=block)); the edges are all-gathered (no device collective: pairs are independent units) and
    added to a fresh graph in the serial order - rank blocks are contiguous and ascending and LinkStage::finalize orders a
    rank's edges by source (link_stage.cpp:123-127), so concatenation in rank order IS the single-process order and
    graph.addEdge draws the same ids.  Returns (merged graph, number of edges this rank linked itself)."""
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    work = make_graph()
    ids = list(work.node_ids)
    lo, hi = source_block(len(ids), rank, world)
    work.link(ctx, node_ids=ids[lo:hi])
    mine = work.edges(with_distances=True)
    work.close()
    parts = [mine]
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, mine, group=group)
    merged = make_graph()
    for part in parts:
        for e in part:
            merged.add_edge(e["source"], e["dest"], e["px"], e["f1"], e["f2"], e["match_index"], e["H"], e["dist"], e["poses"],
                            match_idx=e["match_idx"], is_homography=e["is_homography"])
    return merged, len(mine)


_pinned = {}


def _pinned_bytes(n, tag):
    """A page-locked host staging tensor of at least n bytes, kept between calls (allocation is the expensive part)."""
    import torch

    t = _pinned.get(tag)
    if t is None or t.numel() < n:
        t = torch.empty(max(n, 1) * 5 // 4, dtype=torch.uint8, pin_memory=torch.cuda.is_available())
        _pinned[tag] = t
    return t[:n]


def all_gather_bytes(buf, group=None, stats=None):
    """Every rank's byte buffer (numpy uint8, any length) on every rank: a list indexed by rank.  Backend "nccl": an
    RCCL all-gather of device tensors (xGMI), staged through page-locked host memory on both sides; "gloo": host tensors.
    The slices start at multiples of 8 bytes.  stats (dict): bytes gathered and exchanges are added to it."""
    import torch
    import torch.distributed as dist

    buf = np.ascontiguousarray(buf, np.uint8)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return [buf]
    on_device = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_device else torch.device("cpu")
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([len(buf)], dtype=torch.int64, device=dev), group=group)
    sizes = [int(v) for v in sizes.cpu().tolist()]
    cap = (max(sizes) + 7) // 8 * 8
    if cap == 0:
        return [np.zeros(0, np.uint8) for _ in range(world)]
    if on_device:
        stage = _pinned_bytes(cap, "send")
        stage[:len(buf)] = torch.from_numpy(buf)
        mine = stage.to(dev, non_blocking=True)
        full = torch.empty(world * cap, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(full, mine, group=group)
        back = _pinned_bytes(world * cap, "recv")
        back.copy_(full, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        host = back.numpy()
    else:
        mine = torch.zeros(cap, dtype=torch.uint8)
        mine[:len(buf)] = torch.from_numpy(buf)
        full = torch.empty(world * cap, dtype=torch.uint8)
        dist.all_gather_into_tensor(full, mine, group=group)
        host = full.numpy()
    if stats is not None:
        stats["exchanges"] = stats.get("exchanges", 0) + 1
        stats["bytes_gathered"] = stats.get("bytes_gathered", 0) + sum(sizes)
    # copies: on the RCCL path `host` views a page-locked staging buffer that the next call overwrites
    return [host[r * cap:r * cap + sizes[r]].copy() for r in range(world)]


def survey_sharded(ctx, graph, model, positions, orientations, images_block, width, height, group=None, max_keypoints=30000,
                   on_device=True, edges_to=None):
    """ONE survey's load + link stages over the ranks of `group` (host.Shard / och_shard_*): this rank extracts its
    contiguous block of the images (`images_block`: the block's views) and links the directed pairs the block owns; the
    40 px subsets and then the pairs' results are all-gathered, and every rank finishes with the same graph edges - the
    ones a single process builds, ids included (LinkStage::finalize's sort, link_stage.cpp:123-127).  `graph` holds the
    camera model `model` and no nodes.  edges_to = a rank: only that rank imports the other ranks' edges (the one that
    goes on to relax the survey); the others end with their own edges only.  Returns a dict of the stage's numbers
    (seconds per phase, gathered bytes)."""
    import time

    import torch.distributed as dist

    from .host import Shard

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    sh = Shard(graph, ctx, model, positions, orientations, rank, world)
    stats = {}
    try:
        sh.load_link_local(images_block, width, height, max_keypoints, on_device)
        t_ex = 0.0
        if world > 1:
            t0 = time.perf_counter()
            parts = all_gather_bytes(sh.subsets_export(), group, stats)
            t_ex += time.perf_counter() - t0
            for r, part in enumerate(parts):
                if r != rank:
                    sh.subsets_import(part)
            sh.link_remote()
            t0 = time.perf_counter()
            parts = all_gather_bytes(sh.edges_export(), group, stats)
            t_ex += time.perf_counter() - t0
            for r, part in enumerate(parts):
                if r != rank and edges_to in (None, rank):
                    sh.edges_import(part)
        feats, sparse, link_timers, seconds = sh.finalize()
        out = dict(features=feats, sparse=sparse, link_timers=link_timers, seconds=seconds, exchange_s=t_ex,
                   block=(sh.first, sh.count), rank=rank, world=world, **sh.counts())
        out.update(stats)
        return out
    finally:
        sh.close()
