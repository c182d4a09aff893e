import sys; sys.path.insert(0, "/root/repo")
import time, numpy as np
from opencalibration_amd import capi, host, pipeline, synth
grid = synth.make_grid(seed=1, rows=10, cols=20, feats=64)
ctx = capi.Context(0)
images, shape = pipeline.synthetic_views(ctx, grid, seed=7)
n, h, w = shape
hostimgs = np.stack([ctx.synth_views_read(images, i, w, h) for i in range(n)])
print("host array GB", hostimgs.nbytes / 1e9)
for rep in range(2):
    t0 = time.perf_counter(); f = host.extract_features_batch(ctx, hostimgs, 30000); t1 = time.perf_counter()
    f2 = host.extract_features_batch(ctx, images, 30000, device_shape=(n, h, w)); t2 = time.perf_counter()
    print("extract from host memory: %.3f s (%.0f images/s, %.1f GB/s of pixels); from HBM: %.3f s (%.0f images/s)" % (t1 - t0, n / (t1 - t0), hostimgs.nbytes / 1e9 / (t1 - t0), t2 - t1, n / (t2 - t1)))
