"""Load stage from page-locked host memory (the reference's boundary: host images) for a few chunk sizes / numbers of launch
sequences in flight: images/s and GB/s of pixels.  Usage: python scripts/probe_pcie_ingest.py [n_images]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
import numpy as np  # noqa: E402

from opencalibration_amd import capi, host, pipeline, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = capi.Context(0)
grid = synth.make_grid(seed=12345, rows=10, cols=n // 10, feats=64)
images, shape = pipeline.synthetic_views(ctx, grid, seed=7)
_, h, w = shape
hostviews, release = ctx.host_array((n, h, w, 3))
for i in range(n):
    ctx.synth_views_read_into(images, i, w, h, hostviews[i])


def run():
    g = host.Graph()
    m = g.add_model(grid.model)
    t0 = time.perf_counter()
    g.load_images(ctx, hostviews, m, grid.position[:n], 30000, device_shape=None)
    dt = time.perf_counter() - t0
    g.close()
    return dt


run()
for streams in (4, 5, 3):
    for chunk in (25,):  # (images per upload: fixed at 25 since round 4, extract_features.cpp)
        os.environ["OCHIP_EXTRACT_STREAMS"] = str(streams)
        best = min(run() for _ in range(2))
        print(f"streams {streams} chunk {chunk:3d}: {n / best:7.1f} images/s  {n * h * w * 3 / best / 1e9:5.1f} GB/s", flush=True)
release()
