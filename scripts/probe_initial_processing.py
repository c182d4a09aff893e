"""INITIAL_PROCESSING in the reference's schedule (pipeline.run_initial_processing) on rendered views: seconds per step and
per stage.  usage: probe_initial_processing.py [C3] [batch] [sequential=0]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opencalibration_amd import capi, pipeline, synth

cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sequential = len(sys.argv) > 3 and sys.argv[3] == "1"
grid = synth.make_grid(seed=12345, rows=cfg["rows"], cols=cfg["cols"], feats=64)
ctx = capi.Context(0)
images, shape = pipeline.synthetic_views(ctx, grid, seed=7)
for rep in range(3):
    g, inc = pipeline.run_initial_processing(ctx, grid, images, shape, batch=batch, sequential=sequential)
    err = pipeline.orientation_errors(g.orientations(), grid.orientation)
    print("rep %d: %.3f s -> %.1f images/s; steps %s; load %.3f link %.3f relax %.3f (device %.3f, host set-up %.3f); solves %d iterations %d; "
          "median error %.2e, unoriented %d" % (rep, inc["seconds"], grid.n_images / inc["seconds"], inc["step_seconds"], inc["load_runner_s"],
                                                 inc["link_runner_s"], inc["relax_runner_s"], inc["relax_device_s"], inc["relax_setup_host_s"],
                                                 inc["solves"], inc["lm_iterations"], float(np.median(err)), int(np.sum(~np.isfinite(err)))), flush=True)
    if rep == 2:
        print("per step [init, load runner, link runner, relax runner, finalize]:", inc["step_stage_seconds"], flush=True)
    g.close()
