"""The relax flavours of the later pipeline states on a C-sized survey (synthetic features, device link): ground plane,
then {ORIENTATION, GROUND_MESH} on the 4-vertex minimal mesh (what MESH_REFINEMENT run 0 relaxes, pipeline.cpp:681-707),
then on a camera-spacing grid mesh (the legacy rebuildMesh surface: thousands of vertices in the band of the system).
usage: probe_relax_mesh.py [C2|C3] [feats]"""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from opencalibration_amd import capi, host, pipeline, synth

cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
feats = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["feats"]
grid = synth.make_grid(seed=12345, rows=cfg["rows"], cols=cfg["cols"], feats=feats)
ctx = capi.Context(0)
g = host.Graph.from_synthetic(grid)
start = pipeline.perturbed_orientations(grid, 0.1, 99)
g.set_orientations(start)
t0 = time.perf_counter()
g.link(ctx)
print("link %.3f s, %d edges" % (time.perf_counter() - t0, g.num_edges), flush=True)


def show(name, r, dt):
    err = pipeline.orientation_errors(r["orientation"], grid.orientation)
    keys = ("solves", "iterations_total", "residual_blocks", "track_blocks", "two_ray_blocks", "mesh_vertices", "unknowns",
            "setup_host_s", "device_s", "initial_cost", "final_cost")
    print(name, "%.3f s" % dt, {k: (round(r[k], 4) if isinstance(r[k], float) else r[k]) for k in keys if k in r},
          "median err %.2e rad" % np.median(err), flush=True)


O = host.relax_options
for rep in range(2):
    t0 = time.perf_counter()
    plane = g.relax(ctx, start, O("ORIENTATION", "GROUND_PLANE"))
    show("ground plane    ", plane, time.perf_counter() - t0)
for rep in range(2):
    t0 = time.perf_counter()
    minimal_surface = host.rebuild_mesh(grid.position, plane["surface"], minimal=True)
    m1 = g.relax(ctx, plane["orientation"], O("ORIENTATION", "GROUND_MESH"), 0.1, previous=minimal_surface)
    show("mesh, 4 vertices", m1, time.perf_counter() - t0)
for rep in range(2):
    t0 = time.perf_counter()
    grid_surface = host.rebuild_mesh(grid.position, m1["surface"], minimal=False)
    m2 = g.relax(ctx, m1["orientation"], O("ORIENTATION", "GROUND_MESH"), 0.1, previous=grid_surface)
    show("mesh, grid      ", m2, time.perf_counter() - t0)
for kid, name in ((capi.K_RELAX_EVAL, "eval"), (capi.K_RELAX_SOLVE, "solve")):
    n, ms = ctx.profile_get(kid)
    print(name, n, "launches", round(ms, 2), "ms total", round(ms / max(n, 1), 4), "ms avg")
