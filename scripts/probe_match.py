"""GPU-box probe: Hamming 2-NN kernel time on a C2-shaped slice, host core count, device info."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opencalibration_amd import capi, host, synth  # noqa: E402

rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5, 10)
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
t = time.time()
g = synth.make_grid(rows, cols, feats=4096)
print("grid", g.n_images, "images", len(g.loc), "features", round(time.time() - t, 2), "s")
ctx = capi.Context(0)
print(ctx.device_info())
t = time.time()
subsets = [host.subsample(*g.image(i)[:2], 40.0, int(g.num_sparse[i])) for i in range(g.n_images)]
print("subsample", round(time.time() - t, 2), "s; mean subset", np.mean([len(s) for s in subsets]))
ctx.descriptors_reserve(g.n_images, sum(len(s) for s in subsets))
for i, s in enumerate(subsets):
    ctx.upload_descriptors(i, g.image(i)[2][s.astype(np.int64)])
xy = g.position[:, :2]
d = np.sum((xy[:, None] - xy[None]) ** 2, -1)
knn = np.argsort(d, axis=1, kind="stable")[:, :10]
pairs = np.array([(a, b) for a in range(g.n_images) for b in knn[a] if a != b], capi.PAIR_DTYPE)
n1 = np.array([len(subsets[a]) for a in pairs["image_1"]], np.uint64)
off = np.concatenate([[0], np.cumsum(n1)[:-1]]).astype(np.uint64)
total = int(n1.sum())
compares = float(sum(len(subsets[a]) * len(subsets[b]) for a, b in pairs))
for it in range(4):
    ctx.profile_reset()
    t = time.time()
    out = ctx.match_batch(pairs, off, total)
    wall = time.time() - t
    n, ms = ctx.profile_get(capi.K_MATCH)
    print(f"iter {it}: pairs {len(pairs)} kernel {ms:.3f} ms wall {wall*1e3:.1f} ms  "
          f"{compares/ms/1e6:.1f} Gcompares/s  {compares*35/ms/1e9:.2f} Tlane-op/s (peak ~78.6)")
