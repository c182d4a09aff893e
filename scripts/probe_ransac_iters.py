"""Distribution of RANSAC iterations per directed pair on the image path (C2 grid), and the kernel time."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from opencalibration_amd import capi, host, pipeline, synth

cfg = synth.CONFIGS["C2"]
grid = synth.make_grid(seed=12345, rows=cfg["rows"], cols=cfg["cols"], feats=64)
ctx = capi.Context(0)
images, shape = pipeline.synthetic_views(ctx, grid, seed=7)
n, h, w = shape
g = host.Graph()
m = g.add_model(grid.model)
g.load_images(ctx, images, m, grid.position, device_shape=(n, h, w))
ctx.profile_reset()
t0 = time.perf_counter()
g.link(ctx, keep_debug=True)
print("link s", time.perf_counter() - t0, "ransac kernel", ctx.profile_get(capi.K_RANSAC), "match kernel", ctx.profile_get(capi.K_MATCH))
dbg = g.link_debug()
it = np.array([d["iterations"] for d in dbg]); M = np.array([len(d["i1"]) for d in dbg]); ninl = np.array([int(d["inliers"].sum()) for d in dbg])
print("pairs", len(dbg), "iterations: mean %.1f median %d p90 %d p99 %d max %d" % (it.mean(), np.median(it), np.percentile(it, 90), np.percentile(it, 99), it.max()))
print("matches: mean %.0f  inlier ratio: mean %.2f p10 %.2f min %.2f" % (M.mean(), (ninl / np.maximum(M, 1)).mean(), np.percentile(ninl / np.maximum(M, 1), 10), (ninl / np.maximum(M, 1)).min()))
for lo, hi in [(0, 21), (21, 50), (50, 200), (200, 1000), (1000, 10001)]:
    sel = (it >= lo) & (it < hi)
    print(f"iters [{lo},{hi}): {sel.sum()} pairs, mean M {M[sel].mean() if sel.any() else 0:.0f}, mean ratio {(ninl[sel]/np.maximum(M[sel],1)).mean() if sel.any() else 0:.2f}")
