import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opencalibration_amd import capi, host, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
grid = synth.make_grid(**synth.CONFIGS[cfg])
ctx = capi.Context(0)
rng = np.random.default_rng(99)
axes = rng.normal(size=(grid.n_images, 3)); axes /= np.linalg.norm(axes, axis=1, keepdims=True)
dq = np.concatenate([axes * np.sin(0.05), np.full((grid.n_images, 1), np.cos(0.05))], axis=1)
start = synth.quat_mul(grid.orientation, dq)
g = host.Graph.from_synthetic(grid); g.set_orientations(start)
print(g.link(ctx))
t = time.time(); rel = g.relax_ground_plane(ctx, start); print("relax", time.time() - t)
print({k: v for k, v in rel.items() if k not in ("orientation",)})
dots = np.abs(np.sum(rel["orientation"] * grid.orientation, axis=1))
ang = 2 * np.arccos(np.clip(dots, 0, 1))
print("err: max", ang.max(), "median", np.median(ang), "n>1e-2:", int((ang > 1e-2).sum()), "argmax", int(ang.argmax()))
