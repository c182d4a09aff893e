import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from opencalibration_amd import capi, host, pipeline, synth
grid = synth.make_grid(seed=12345, rows=2, cols=5, feats=64)
ctx = capi.Context(0)
images, shape = pipeline.synthetic_views(ctx, grid, seed=7)
n, h, w = shape
feats = host.extract_features_batch(ctx, images, 30000, device_shape=(n, h, w))
for loc, st, de, ns in feats[:4]:
    idx = host.subsample(loc, st, 40.0, ns)
    print(len(st), ns, len(idx))
