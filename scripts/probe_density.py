import sys, os, time
import numpy as np
sys.path.insert(0, ".")
from opencalibration_amd import capi, host, pipeline, synth
grid = synth.make_grid(seed=12345, rows=4, cols=5, feats=64)
ctx = capi.Context(0)
for sp in ("21", "18", "16", "14"):
    os.environ["OCHIP_BLOB_SPACING"] = sp
    images, shape = pipeline.synthetic_views(ctx, grid, seed=7)
    start = pipeline.perturbed_orientations(grid, 0.1, 99)
    g, res, t = pipeline.run(ctx, grid, images, shape, start, overlap=False)
    dbg_n = [e["n_matches"] for e in g.edges()]
    # subset size: number of features entering the matcher ~ via link debug? use och subsample on node features
    n, h, w = shape
    feats = host.extract_features_batch(ctx, images, 30000, device_shape=(4, h, w))
    subs = [len(host.subsample(f[0], f[1], 40.0, f[3])) if hasattr(host, "subsample") else -1 for f in feats]
    print(sp, "features/img", res["features_per_image"], "sparse", res["sparse_per_image"], "subset", subs, "matches/edge", np.mean(dbg_n), "inliers", np.mean([e["n_inliers"] for e in g.edges()]), "t", {k: round(v,3) for k,v in t.items()}, flush=True)
    g.close()
    ctx.synth_views_free(images)
