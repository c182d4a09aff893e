"""Worker of tests/test_gpu_survey_sharded.py: 2 ranks (gloo rendezvous) sharing cuda:0 run ONE survey from pixels -
extraction by image block, the directed pairs of a block linked by its rank, the 40 px subsets and the pairs' results
all-gathered (parallel.survey_sharded over och_shard_*) - and the relax sharded over the ranks.  Every rank must end with the
edge list a single process builds (payload for payload, id for id), its own block's feature lists, and the relaxed
orientations of the single-process run, bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opencalibration_amd import capi, host, parallel, pipeline, synth  # noqa: E402


def signature(g):
    out = []
    for e in g.edges(with_distances=True):
        out.append((e["source"], e["dest"], e["n_matches"], e["n_inliers"], e["H"].tobytes(), e["f1"].tobytes(), e["f2"].tobytes(),
                    e["match_index"].tobytes(), e["px"].tobytes(), e["poses"].tobytes(), e["dist"].tobytes(),
                    e["match_idx"].tobytes(), e["is_homography"]))
    return out


def main():
    rows, cols = (int(v) for v in os.environ.get("SHARD_TEST_GRID", "4x6").split("x"))
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ctx = capi.Context(0)
    grid = synth.make_grid(rows, cols, feats=64, seed=31)
    images, shape = pipeline.synthetic_views(ctx, grid, seed=5)
    n, h, w = shape
    start = pipeline.perturbed_orientations(grid, 0.05, 3)
    single, res, _ = pipeline.run(ctx, grid, images, shape, start)            # the single-process survey
    ref_sig = signature(single)

    lo, cnt = host.shard_block(n, rank, world)
    g = host.Graph()
    mid = g.add_model(grid.model)
    st = parallel.survey_sharded(ctx, g, mid, grid.position, start, images + lo * h * w * 3, w, h)
    checks = {"edges": g.num_edges == single.num_edges and signature(g) == ref_sig,
              "ids": g.node_ids == single.node_ids,
              "pairs": st["pairs_in_block"] > 0 and st["pairs_across_blocks"] > 0 and st["halo_images"] > 0}
    # the block's images carry their features, the others none
    ta, tb = single.node_table(), g.node_table()
    own = np.zeros(n, bool)
    own[lo:lo + cnt] = True
    checks["features"] = bool(np.array_equal(ta["features"][own], tb["features"][own]) and not tb["features"][~own].any()
                              and np.array_equal(ta["sparse"][own], tb["sparse"][own]))
    pa, pb = single.node_payload(lo), g.node_payload(lo)
    checks["payload"] = all(np.array_equal(pa[k], pb[k]) for k in ("loc", "strength", "desc"))
    # relax: sharded over the ranks, bit-identical to the single-process relax
    exch = parallel.relax_exchange()
    rel = g.relax_ground_plane(ctx, start, shard=(rank, world, exch))
    checks["relax"] = bool(np.array_equal(rel["orientation"], res["relax"]["orientation"]))
    # edges_to: only the rank that goes on to relax the survey imports the others' edges
    g2 = host.Graph()
    m2 = g2.add_model(grid.model)
    parallel.survey_sharded(ctx, g2, m2, grid.position, start, images + lo * h * w * 3, w, h, edges_to=1)
    checks["edges_to"] = (g2.num_edges == single.num_edges and signature(g2) == ref_sig) if rank == 1 else \
        (0 < g2.num_edges < single.num_edges)
    flags = [None] * world
    dist.all_gather_object(flags, (checks, st["bytes_gathered"], st["exchanges"], int(single.num_edges)))
    if rank == 0:
        ok = all(all(f[0].values()) for f in flags) and flags[0][2] == 2 and flags[0][3] > 5 * n
        print("SHARDED_SURVEY", "OK" if ok else "MISMATCH", flags, flush=True)
    for x in (single, g, g2):
        x.close()
    ctx.synth_views_free(images)
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
