"""Worker of tests/test_gpu_link_sharded.py: 2 ranks (gloo rendezvous) sharing cuda:0.  One survey, the link stage sharded
by source-image block over the ranks on the DEVICE path (parallel.link_sharded -> LinkStage on libochip), edges gathered;
the merged graph must equal the graph one process links on its own, payload for payload and id for id."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opencalibration_amd import capi, host, parallel, synth  # noqa: E402


def signature(g):
    out = []
    for e in g.edges(with_distances=True):
        out.append((e["source"], e["dest"], e["n_matches"], e["n_inliers"], e["H"].tobytes(), e["f1"].tobytes(), e["f2"].tobytes(),
                    e["match_index"].tobytes(), e["px"].tobytes(), e["poses"].tobytes(), e["dist"].tobytes(),
                    e["match_idx"].tobytes(), e["is_homography"]))
    return out


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ctx = capi.Context(0)
    grid = synth.make_grid(4, 6, feats=700, seed=23)
    make = lambda: host.Graph.from_synthetic(grid)
    single = make()
    single.link(ctx)
    merged, n_mine = parallel.link_sharded(make, ctx)
    same = signature(single) == signature(merged) and single.num_edges == merged.num_edges > 100
    # and the relax of the merged graph is the relax of the single-process graph
    rng = np.random.default_rng(3)
    axes = rng.normal(size=(grid.n_images, 3))
    axes /= np.linalg.norm(axes, axis=1, keepdims=True)
    start = synth.quat_mul(grid.orientation, np.concatenate([axes * np.sin(0.05), np.full((grid.n_images, 1), np.cos(0.05))], 1))
    a = single.relax_ground_plane(ctx, start)
    b = merged.relax_ground_plane(ctx, start)
    same = same and np.array_equal(a["orientation"], b["orientation"])
    flags = [None] * world
    dist.all_gather_object(flags, (bool(same), int(n_mine), int(single.num_edges)))
    if rank == 0:
        ok = all(f[0] for f in flags) and sum(f[1] for f in flags) == flags[0][2] and all(0 < f[1] < f[2] for f in flags)
        print("SHARDED_LINK", "OK" if ok else "MISMATCH", flags, flush=True)
    single.close()
    merged.close()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
