#!/usr/bin/env python3
"""Where the host part of the ground-plane relax goes: one C3 survey (load + link + relax) with OCHIP_VERBOSE=relax
(lap times of RelaxProblem::setup on stderr).  usage: [taskset -c 0,1] probe_relax_setup.py [C2|C3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OCHIP_BLOB_SPACING", "16")
os.environ["OCHIP_VERBOSE"] = "relax"
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
from opencalibration_amd import capi, pipeline, synth

cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
grid = synth.make_grid(seed=12345, rows=cfg["rows"], cols=cfg["cols"], feats=64)
ctx = capi.Context(0)
images, shape = pipeline.synthetic_views(ctx, grid)
start = pipeline.perturbed_orientations(grid, 0.1, 4)
for rep in range(2):
    g, res, t = pipeline.run(ctx, grid, images, shape, start)
    print("rep", rep, {k: round(v, 4) for k, v in t.items()}, "setup_host_s", res["relax"]["setup_host_s"], file=sys.stderr)
    g.close()
