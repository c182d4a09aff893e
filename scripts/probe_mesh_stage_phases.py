#!/usr/bin/env python3
"""The mesh-flavour relax stage of the bench's survey (from pixels) with the set-up's phase lines summed over its groups.
usage: OCHIP_VERBOSE=relax probe_mesh_stage_phases.py [C3] 2> log; the script re-reads its own stderr file when given as argv[2]."""
import collections
import os
import re
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarise(path):
    tot, cnt = collections.OrderedDict(), collections.Counter()
    for line in open(path, errors="replace"):
        m = re.match(r"\[(relax mesh setup|relax mesh tracks|relax setup)\]\s+(.*?)\s+([0-9.]+) ms", line)
        if m:
            k = m.group(1).replace("relax ", "") + ": " + m.group(2)
            tot[k] = tot.get(k, 0.0) + float(m.group(3))
            cnt[k] += 1
    for k, v in tot.items():
        print("%-55s %4d x  %9.2f ms summed" % (k, cnt[k], v))


if len(sys.argv) > 2:
    summarise(sys.argv[2])
    sys.exit(0)

import numpy as np
from opencalibration_amd import capi, host, pipeline, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
grid = synth.make_grid(**synth.CONFIGS[cfg])
ctx = capi.Context(0)
images, shape = pipeline.synthetic_views(ctx, grid, seed=11)
start = pipeline.perturbed_orientations(grid, 0.1, 4)
os.environ.pop("OCHIP_VERBOSE", None)
verbose = os.environ.pop("OCHIP_PROBE_VERBOSE", "relax")
gg, _, _ = pipeline.run(ctx, grid, images, shape, start)
plane = gg.relax(ctx, start, host.relax_options("ORIENTATION", "GROUND_PLANE"))
seed_mesh = host.rebuild_mesh(grid.position, plane["surface"], minimal=True)
for rep in range(3):
    if rep == 2:
        os.environ["OCHIP_VERBOSE"] = verbose
        print("==== verbose repetition", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    ms = gg.relax_stage(ctx, host.relax_options("ORIENTATION", "GROUND_MESH"), 0.1, previous=seed_mesh)
    print("rep %d: %.3f s, groups %d, host set-up summed %.3f, device summed %.3f, blocks %d (tracks %d, two-ray %d)"
          % (rep, time.perf_counter() - t0, ms["groups"], ms["setup_host_s"], ms["device_s"], ms["residual_blocks"], ms["track_blocks"],
             ms["two_ray_blocks"]), flush=True)
