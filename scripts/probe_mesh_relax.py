"""The single-group GROUND_MESH relax of a C3-sized survey (1 000 cameras, ~650 k residual blocks): device seconds per LM
iteration."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opencalibration_amd import capi, host, pipeline, synth  # noqa: E402

ctx = capi.Context(0)
grid = synth.make_grid(seed=2025, feats=2048, **{k: v for k, v in synth.CONFIGS["C3"].items() if k != "feats"})
g = host.Graph.from_synthetic(grid)
start = pipeline.perturbed_orientations(grid, 0.1, 7)
g.set_orientations(start)
g.link(ctx)
plane = g.relax(ctx, start, host.relax_options("ORIENTATION", "GROUND_PLANE"))
seed = host.rebuild_mesh(grid.position, plane["surface"], minimal=True)
O = host.relax_options("ORIENTATION", "GROUND_MESH")
for rep in range(3):
    g.set_orientations(plane["orientation"])
    t0 = time.perf_counter()
    r = g.relax(ctx, plane["orientation"], O, 0.1, previous=seed)
    dt = time.perf_counter() - t0
    print(f"mesh relax: blocks {int(r['residual_blocks'])} unknowns {int(r['unknowns'])} iterations {int(r['iterations_total'])} "
          f"device {r['device_s']:.4f} s ({1e3 * r['device_s'] / max(r['iterations_total'], 1):.2f} ms / iteration) wall {dt:.3f} s", flush=True)
