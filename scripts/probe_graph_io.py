import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from opencalibration_amd import capi, host, synth
grid = synth.make_grid(**synth.CONFIGS["C1"])
ctx = capi.Context(0)
g = host.Graph.from_synthetic(grid)
g.link(ctx)
start = grid.orientation.copy()
g.set_orientations(start)
text = g.to_json()
print("json bytes", len(text), flush=True)
g2 = host.Graph().from_json(text)
print("loaded", g2.num_nodes, g2.num_edges, flush=True)
order = [g.node_ids.index(i) for i in g2.node_ids]
a = g.relax_ground_plane(ctx, start)
print("relax a", a["residual_blocks"], flush=True)
b = g2.relax_ground_plane(ctx, start[order])
print("relax b", b["residual_blocks"], flush=True)
