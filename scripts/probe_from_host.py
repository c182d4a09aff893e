"""The whole step from page-locked host memory, two surveys in flight (bench.py's pcie_inclusive leg on its own).
usage: probe_from_host.py [n_images=1000]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
import numpy as np  # noqa: E402

from opencalibration_amd import capi, host, pipeline, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cfg = synth.CONFIGS["C3"]
grid = synth.make_grid(seed=12345, rows=cfg["rows"], cols=cfg["cols"], feats=64)
ctx = capi.Context(0)
images, shape = pipeline.synthetic_views(ctx, grid, seed=7)
_, h, w = shape
hostviews, release = ctx.host_array((n, h, w, 3))
for i in range(n):
    ctx.synth_views_read_into(images, i, w, h, hostviews[i])
ctx.synth_views_free(images)
start = pipeline.perturbed_orientations(grid, 0.1, 99)
rctx = ctx.sibling(12)
rctx.set_priority(True)
ctx_b = capi.Context(0)
lock, threads = threading.Lock(), []


def relax_locked(g, res, t):
    with lock:
        pipeline.relax_step(rctx, g, start, res, t)
    g.close()


def lane(c, steps):
    for _ in range(steps):
        gs, rs, ts = pipeline.run(c, grid, None, (n, h, w), start, host_images=hostviews, relax=False)
        th = threading.Thread(target=relax_locked, args=(gs, rs, ts))
        th.start()
        threads.append(th)


for c in (ctx, ctx_b):
    pipeline.run(c, grid, None, (n, h, w), start, host_images=hostviews, relax=False)[0].close()
for rep in range(2):
    per_lane = 4
    t0 = time.perf_counter()
    lanes = [threading.Thread(target=lane, args=(c, per_lane)) for c in (ctx, ctx_b)]
    lanes[0].start()
    time.sleep(0.3)
    lanes[1].start()
    for th in lanes:
        th.join()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    print("two surveys in flight: %.1f images/s (%.1f GB/s of pixels)" % (2 * per_lane * n / dt, 2 * per_lane * n * h * w * 3 / dt / 1e9), flush=True)
release()
