"""Worker of tests/test_gpu_relax_sharded_c3.py: 2 ranks (gloo) sharing cuda:0, a survey of BASELINE's C3 / C4 size (1 000
cameras, 25 x 40 grid).  Every rank links the same graph, then three relax problems are solved once by one process and
once over the two ranks; the results must be equal to the bit on every rank:
  plane    the single global group {ORIENTATION, GROUND_PLANE}: residual blocks over the ranks, records exchanged
  mesh     the single global group {ORIENTATION, GROUND_MESH} of FINAL_GLOBAL_RELAX's last run (pipeline.cpp:645-664): the
           general engine's evaluation over the ranks (ochip_relaxg_desc.shard_world, ochip_relaxg_set_exchange)
  groups   the clustered stage (floor(n / 50) = 20 groups, relax_stage.cpp:49-57): groups over the ranks, no exchange
           inside a solve, results gathered, mergeSurfaceModels over all groups on every rank"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opencalibration_amd import capi, host, parallel, pipeline, synth  # noqa: E402


def surfaces_equal(a, b):
    x, y = a.arrays(), b.arrays()
    return all(np.array_equal(x[k], y[k]) for k in ("vertices", "edges", "cloud"))


def main():
    rows, cols = (int(v) for v in os.environ.get("SHARD_TEST_GRID", "25x40").split("x"))
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ctx = capi.Context(0)
    grid = synth.make_grid(rows, cols, feats=int(os.environ.get("SHARD_TEST_FEATS", "1024")), seed=77)
    g = host.Graph.from_synthetic(grid)
    start = pipeline.perturbed_orientations(grid, 0.1, 5)
    g.set_orientations(start)
    g.link(ctx)
    exch = parallel.relax_exchange()
    checks, info = {}, {}
    # ---- plane, one group
    ref = g.relax_ground_plane(ctx, start)
    g.set_orientations(start)
    got = g.relax_ground_plane(ctx, start, shard=(rank, world, exch))
    checks["plane"] = bool(np.array_equal(ref["orientation"], got["orientation"]) and np.array_equal(ref["plane"], got["plane"])
                           and ref["iterations_total"] == got["iterations_total"])
    info["plane_blocks"] = int(got["residual_blocks"])
    # ---- mesh, one group: minimal mesh seeded from the plane (pipeline.cpp:681-707)
    O = host.relax_options("ORIENTATION", "GROUND_MESH")
    ori0 = got["orientation"]
    plane = g.relax(ctx, ori0, host.relax_options("ORIENTATION", "GROUND_PLANE"))
    seed = host.rebuild_mesh(grid.position, plane["surface"], minimal=True)
    g.set_orientations(ori0)
    mref = g.relax(ctx, ori0, O, 0.1, previous=seed)
    g.set_orientations(ori0)
    mgot = g.relax(ctx, ori0, O, 0.1, previous=seed, shard=(rank, world, exch))
    checks["mesh"] = bool(np.array_equal(mref["orientation"], mgot["orientation"]) and surfaces_equal(mref["surface"], mgot["surface"])
                          and mref["iterations_total"] == mgot["iterations_total"] and mref["final_cost"] == mgot["final_cost"])
    info["mesh_blocks"] = int(mgot["residual_blocks"])
    info["mesh_err"] = float(np.median(pipeline.orientation_errors(mgot["orientation"], grid.orientation)))
    # ---- clustered stage: groups over the ranks
    g.set_orientations(ori0)
    sref = g.relax_stage(ctx, O, 0.1, previous=seed)
    oref = g.orientations().copy()
    g.set_orientations(ori0)
    sgot = g.relax_stage(ctx, O, 0.1, previous=seed, shard=(rank, world, parallel.all_gather_bytes))
    checks["groups"] = bool(np.array_equal(oref, g.orientations()) and surfaces_equal(sref["surface"], sgot["surface"])
                            and sref["groups"] == sgot["groups"] and np.array_equal(sref["group_of_node"], sgot["group_of_node"])
                            and sref["iterations_total"] == sgot["iterations_total"])
    info["groups"] = int(sgot["groups"])
    flags = [None] * world
    dist.all_gather_object(flags, (checks, info))
    if rank == 0:
        ok = all(all(f[0].values()) for f in flags) and info["groups"] == grid.n_images // 50 and info["mesh_err"] < 2e-3
        print("SHARDED_RELAX_C3", "OK" if ok else "MISMATCH", flags, flush=True)
    g.close()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
