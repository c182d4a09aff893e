"""LM iterations per second of the C3 ground-plane relax (3 003 unknowns) alone on the device, from synthetic features
(no extraction): A/B of the camera-graph dissection (OCHIP_TEST_HOOKS=no_dissect, OCHIP_RELAX_DISSECT_G) - one process per
setting, the knobs are read once.  usage: python scripts/probe_relax_dissect.py [C3 | ROWSxCOLS[xFEATURES]] [repeats]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opencalibration_amd import capi, host, synth

config = sys.argv[1] if len(sys.argv) > 1 else "C3"
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if "x" in config:  # rows x cols (x features per image)
    dims = [int(v) for v in config.split("x")]
    grid = synth.make_grid(seed=5, rows=dims[0], cols=dims[1], feats=dims[2] if len(dims) > 2 else 512)
else:
    grid = synth.make_grid(**synth.CONFIGS[config])
ctx = capi.Context(0)
g = host.Graph.from_synthetic(grid)
g.link(ctx)
rng = np.random.default_rng(1)
start = grid.orientation.copy()
from opencalibration_amd import pipeline
start = pipeline.perturbed_orientations(grid, 0.1, 4)
rates = []
for r in range(repeats):
    g.set_orientations(start)
    res = g.relax_ground_plane(ctx, start)
    it, dev = int(res["iterations_total"]), float(res["device_s"])
    rates.append(it / dev)
    mem = ctx.relax_memory()
print("%s: %d iterations, device %.4f s, LM it/s %s, system %s" % (config, it, dev, ["%.0f" % x for x in rates], mem), flush=True)
print("final_cost %.17g orientation_checksum %.17g" % (float(res["final_cost"]), float(np.abs(res["orientation"]).sum())), flush=True)
g.close()
ctx.close()
