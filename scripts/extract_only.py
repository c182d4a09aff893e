#!/usr/bin/env python3
"""The extract stage alone on the first N cameras of the C2 grid (views rendered into HBM first): the quick workload
for per-kernel traces and counter passes of the extract kernels.  usage: extract_only.py [N=100] [repeats=2]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OCHIP_BLOB_SPACING", "16")
import numpy as np

from opencalibration_amd import capi, host, pipeline, synth


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cfg = synth.CONFIGS["C2"]
    grid = synth.make_grid(seed=12345, rows=cfg["rows"], cols=cfg["cols"], feats=64)   # poses + camera model, as bench.py
    ctx = capi.Context(0)
    images, shape = pipeline.synthetic_views(ctx, grid, block=(0, n))
    for r in range(repeats):
        g = host.Graph()
        mid = g.add_model(grid.model)
        t0 = time.perf_counter()
        feats, sparse = g.load_images(ctx, images, mid, grid.position[:n], 30000, device_shape=shape)
        dt = time.perf_counter() - t0
        print(f"repeat {r}: {n} images in {dt * 1e3:.1f} ms ({dt / n * 1e6:.1f} us/image), {feats:.0f} features/image, {sparse:.0f} sparse")
        g.close()
    ctx.synth_views_free(images)
    ctx.close()


if __name__ == "__main__":
    main()
