"""Where INITIAL_PROCESSING's relax goes, batch by batch (pipeline.run_incremental's schedule on rendered views).
usage: probe_incremental.py [C3] [batch] [verbose_batch] [one_group=1]      verbose_batch: OCHIP_VERBOSE=relax for that batch only"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opencalibration_amd import capi, host, pipeline, synth

cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 100
verbose_batch = int(sys.argv[3]) if len(sys.argv) > 3 else -1
one_group = (sys.argv[4] == "1") if len(sys.argv) > 4 else True   # RelaxStage::init(..., disable_parallelism = true), pipeline.cpp:545
grid = synth.make_grid(seed=12345, rows=cfg["rows"], cols=cfg["cols"], feats=64)
ctx = capi.Context(0)
images, shape = pipeline.synthetic_views(ctx, grid, seed=7)
n, h, w = shape
opts = host.relax_options("ORIENTATION", "GROUND_PLANE")
for rep in range(2):
    g = host.Graph()
    mid = g.add_model(grid.model)
    tot_ll = tot_rx = 0.0
    for b, lo in enumerate(range(0, n, batch)):
        cnt = min(batch, n - lo)
        t0 = time.perf_counter()
        g.load_link_images(ctx, images + lo * h * w * 3, mid, grid.position[lo:lo + cnt], np.full((cnt, 4), np.nan), 30000,
                           device_shape=(cnt, h, w))
        t1 = time.perf_counter()
        if rep == 1 and b == verbose_batch:
            os.environ["OCHIP_VERBOSE"] = "relax"
        st = g.relax_stage(ctx, opts, node_ids=g.node_ids[lo:lo + cnt], disable_parallelism=one_group)
        ctx.synchronize()
        os.environ.pop("OCHIP_VERBOSE", None)
        t2 = time.perf_counter()
        tot_ll += t1 - t0
        tot_rx += t2 - t1
        print("rep %d batch %2d: load+link %.3f s, relax %.3f s (host set-up %.3f, device %.3f), solves %d, iterations %d, last blocks %d, edges %d"
              % (rep, b, t1 - t0, t2 - t1, st["setup_host_s"], st["device_s"], st["solves"], st["iterations_total"],
                 st["residual_blocks"], g.num_edges), flush=True)
    err = pipeline.orientation_errors(g.orientations(), grid.orientation)
    print("rep %d: load+link %.3f s, relax %.3f s, %d images -> %.1f images/s; median error %.2e" %
          (rep, tot_ll, tot_rx, n, n / (tot_ll + tot_rx), float(np.median(err))), flush=True)
    g.close()
