"""Worker of tests/test_gpu_world8.py: EIGHT ranks (gloo rendezvous) sharing cuda:0 - the rank count of the node BASELINE's
scaling curve is quoted on.  One survey from pixels over the ranks (3 x 8 grid: three images per rank; extraction by block,
links by owning rank, two all-gathers), its plane relax and its ground-mesh relax sharded over the 8 ranks, then - on a
400-camera graph linked from synthetic features - the clustered stage with its 8 groups dealt over the ranks.  Every rank
must end with what ONE process computes, bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opencalibration_amd import capi, host, parallel, pipeline, synth  # noqa: E402
from sharded_survey_worker import signature  # noqa: E402
from sharded_relax_c3_worker import surfaces_equal  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ctx = capi.Context(0)
    checks, info = {}, {}
    # ---- one survey from pixels over the ranks
    grid = synth.make_grid(3, 8, feats=64, seed=41)
    images, shape = pipeline.synthetic_views(ctx, grid, seed=9)
    n, h, w = shape
    start = pipeline.perturbed_orientations(grid, 0.05, 3)
    single, res, _ = pipeline.run(ctx, grid, images, shape, start)
    lo, cnt = host.shard_block(n, rank, world)
    g = host.Graph()
    mid = g.add_model(grid.model)
    st = parallel.survey_sharded(ctx, g, mid, grid.position, start, images + lo * h * w * 3, w, h)
    checks["edges"] = g.num_edges == single.num_edges and signature(g) == signature(single) and g.node_ids == single.node_ids
    checks["block"] = cnt == 3 and st["pairs_across_blocks"] > 0
    exch = parallel.relax_exchange()
    rel = g.relax_ground_plane(ctx, start, shard=(rank, world, exch))
    checks["plane"] = bool(np.array_equal(rel["orientation"], res["relax"]["orientation"]))
    O = host.relax_options("ORIENTATION", "GROUND_MESH")
    ori0 = rel["orientation"]
    plane = single.relax(ctx, ori0, host.relax_options("ORIENTATION", "GROUND_PLANE"))
    seed = host.rebuild_mesh(grid.position, plane["surface"], minimal=True)
    # (the mesh flavour's set-up reads the nodes' feature lists, which the sharded survey leaves on the owning rank only: it
    # runs on the graph every rank holds in full)
    single.set_orientations(ori0)
    mref = single.relax(ctx, ori0, O, 0.1, previous=seed)
    single.set_orientations(ori0)
    mgot = single.relax(ctx, ori0, O, 0.1, previous=seed, shard=(rank, world, exch))
    checks["mesh"] = bool(np.array_equal(mref["orientation"], mgot["orientation"]) and surfaces_equal(mref["surface"], mgot["surface"])
                          and mref["iterations_total"] == mgot["iterations_total"])
    info["edges"] = int(g.num_edges)
    single.close(), g.close()
    ctx.synth_views_free(images)
    # ---- the clustered stage: 400 cameras, 8 groups over the 8 ranks
    grid2 = synth.make_grid(10, 40, feats=512, seed=78)
    g2 = host.Graph.from_synthetic(grid2)
    start2 = pipeline.perturbed_orientations(grid2, 0.1, 5)
    g2.set_orientations(start2)
    g2.link(ctx)
    plane2 = g2.relax(ctx, start2, host.relax_options("ORIENTATION", "GROUND_PLANE"))
    seed2 = host.rebuild_mesh(grid2.position, plane2["surface"], minimal=True)
    ori2 = plane2["orientation"]
    g2.set_orientations(ori2)
    sref = g2.relax_stage(ctx, O, 0.1, previous=seed2)
    oref = g2.orientations().copy()
    g2.set_orientations(ori2)
    sgot = g2.relax_stage(ctx, O, 0.1, previous=seed2, shard=(rank, world, parallel.all_gather_bytes))
    checks["groups"] = bool(np.array_equal(oref, g2.orientations()) and surfaces_equal(sref["surface"], sgot["surface"])
                            and sref["groups"] == sgot["groups"] == 8 and np.array_equal(sref["group_of_node"], sgot["group_of_node"])
                            and sref["iterations_total"] == sgot["iterations_total"])
    info["groups"] = int(sgot["groups"])
    g2.close()
    flags = [None] * world
    dist.all_gather_object(flags, (checks, info))
    if rank == 0:
        ok = world == 8 and all(all(f[0].values()) for f in flags)
        print("WORLD8", "OK" if ok else "MISMATCH", flags, flush=True)
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
