import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from opencalibration_amd import capi, host
from oracle import pyoracle as O
from relax_fixtures import *
rows, cols = int(sys.argv[1]), int(sys.argv[2])
ori, pos, edges, model = camera_grid(rows, cols, seed=7)
rng = np.random.default_rng(0)
noisy = np.array([qmul(q, axis_angle(rng.normal(size=3) / 1.7, 0.1)) for q in ori])
exp = O.relax_ground_plane(pos, ori, model, np.arange(len(ori)), noisy, edges)
ctx = capi.Context(0)
got = host.relax_ground_plane(ctx, pos, ori, model, np.arange(len(ori)), noisy, O.pack_edges(edges))
print({k: v for k, v in exp.items() if k not in ("orientation", "plane")})
print({k: v for k, v in got.items() if k not in ("orientation", "plane")})
print(max(qangle(exp["orientation"][i], got["orientation"][i]) for i in range(len(ori))))
