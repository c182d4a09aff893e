"""Dense guided matching on the bench workload: images -> extract -> link -> relax, then densifyMesh against the ground."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opencalibration_amd import capi, host, pipeline, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
grid = synth.make_grid(**synth.CONFIGS[cfg])
ctx = capi.Context(0)
images, shape = pipeline.synthetic_views(ctx, grid, seed=11)
start = pipeline.perturbed_orientations(grid, 0.1, 4)
g, res, t = pipeline.run(ctx, grid, images, shape, start)
print("pipeline", {k: round(v, 3) for k, v in t.items()}, "features/image", res["features_per_image"], "sparse", res["sparse_per_image"])
s = host.rebuild_mesh(grid.position, minimal=True)
a = s.arrays()
v = a["vertices"].copy()
v[:, 2] = grid.plane[0] * v[:, 0] + grid.plane[1] * v[:, 1]
s.set(v, a["edges"])
for rep in range(int(os.environ.get("OCHIP_PROBE_REPS", "2"))):
    s2 = host.Surface().set(v, a["edges"])
    t0 = time.perf_counter()
    st = g.densify_mesh(ctx, s2)
    print("densify %.3f s" % (time.perf_counter() - t0), st)
pts = s2.clouds()[-1]
dz = pts[:, 2] - (grid.plane[0] * pts[:, 0] + grid.plane[1] * pts[:, 1])
print("points", len(pts), "median |dz| %.3f m" % np.median(np.abs(dz)))
