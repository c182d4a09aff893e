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
