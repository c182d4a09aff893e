"""Where the mesh-flavour relax stage of a 1 000-camera survey goes: RelaxStage::init (spectral partition), the group
runners, finalize (merge).  usage: probe_relax_stage.py [C3] [feats]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opencalibration_amd import capi, host, pipeline, synth

cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
feats = int(sys.argv[2]) if len(sys.argv) > 2 else 512
grid = synth.make_grid(seed=12345, rows=cfg["rows"], cols=cfg["cols"], feats=feats)
ctx = capi.Context(0)
g = host.Graph.from_synthetic(grid)
start = pipeline.perturbed_orientations(grid, 0.1, 99)
g.set_orientations(start)
g.link(ctx)
plane = g.relax(ctx, start, host.relax_options("ORIENTATION", "GROUND_PLANE"))
seed = host.rebuild_mesh(grid.position, plane["surface"], minimal=True)
opts = host.relax_options("ORIENTATION", "GROUND_MESH")
L = g.L
for rep in range(3):
    groups = np.full(max(g.num_nodes, 1), -1, np.int64)
    summary = np.zeros(13)
    surface = host.Surface()
    t0 = time.perf_counter()
    st = L.och_relax_stage_begin(g.h, None, 0, 1, 0, opts, 0.1, 0, seed.h, groups)
    t1 = time.perf_counter()
    L.och_relax_stage_run_groups(st, ctx.h, 0, 1)
    t2 = time.perf_counter()
    rc = L.och_relax_stage_end(st, surface.h, summary)
    t3 = time.perf_counter()
    print("rep %d: begin (partition) %.3f s, run groups %.3f s, end (merge) %.3f s; rc %d, groups %d, host set-up summed %.3f, device summed %.3f"
          % (rep, t1 - t0, t2 - t1, t3 - t2, rc, int(summary[12]), summary[6] if len(summary) > 6 else -1, summary[7] if len(summary) > 7 else -1), flush=True)
for kid, name in ((capi.K_RELAX_EVAL, "eval"), (capi.K_RELAX_SOLVE, "solve")):
    n, ms = ctx.profile_get(kid)
    print(name, n, "profiled intervals", round(ms, 2), "ms total", round(ms / max(n, 1), 4), "ms avg")
