import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opencalibration_amd import capi, host, synth, pipeline
rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 5)
grid = synth.make_grid(rows, cols, feats=64)     # only poses / model are used; features come from the images
ctx = capi.Context(0)
t = time.time(); ptr, shape = pipeline.synthetic_views(ctx, grid); print("render", shape, round(time.time() - t, 2), "s")
start = pipeline.perturbed_orientations(grid)
for it in range(2):
    g, res, t = pipeline.run(ctx, grid, ptr, shape, start)
    err = pipeline.orientation_errors(res["relax"]["orientation"], grid.orientation)
    print({k: round(v, 4) for k, v in t.items()}, "features/img", res["features_per_image"], "sparse", res["sparse_per_image"], "edges", res["edges"])
    print("link", {k: round(v, 4) for k, v in res["link_timers"].items()})
    print("relax", {k: v for k, v in res["relax"].items() if k not in ("orientation", "plane")})
    print("err median", np.median(err), "max", err.max())
    e = g.edges(); print("edges with inliers:", sum(1 for x in e if x["n_inliers"] > 0), "mean inliers", np.mean([x["n_inliers"] for x in e]), "mean matches", np.mean([x["n_matches"] for x in e]))
    g.close()
