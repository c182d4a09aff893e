"""Throughput of the whole path on a larger survey than C3 (rows x cols given on the command line): does the rate hold?"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from opencalibration_amd import capi, host, pipeline, synth

rows, cols = int(sys.argv[1]), int(sys.argv[2])
grid = synth.make_grid(seed=12345, rows=rows, cols=cols, feats=64)
ctx = capi.Context(0)
images, shape = pipeline.synthetic_views(ctx, grid, seed=7)
start = pipeline.perturbed_orientations(grid, 0.1, 99)
for step in range(2):
    t0 = time.perf_counter()
    g, res, t = pipeline.run(ctx, grid, images, shape, start, overlap=True)
    dt = time.perf_counter() - t0
    err = float(np.median(pipeline.orientation_errors(res["relax"]["orientation"], grid.orientation)))
    print("step %d: %d images in %.3f s = %.0f images/s; stages %s; edges %d; relax unknowns %d, iterations %d, median orientation error %.2e rad" % (
        step, grid.n_images, dt, grid.n_images / dt, {k: round(v, 3) for k, v in t.items()}, res["edges"],
        3 * grid.n_images + 3, int(res["relax"]["iterations_total"]), err), flush=True)
    g.close()
