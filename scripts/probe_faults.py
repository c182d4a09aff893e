"""Minor page faults per step of the load + link stages (are the feature vectors' pages faulted in again every survey?)."""
import os
import resource
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
import ctypes  # noqa: E402

if os.environ.get("KEEP_HEAP", "1") != "0":
    libc = ctypes.CDLL("libc.so.6")
    libc.mallopt(-3, 32 << 20), libc.mallopt(-1, 1 << 30), libc.mallopt(-2, 64 << 20)   # M_MMAP_THRESHOLD, M_TRIM_THRESHOLD, M_TOP_PAD
from opencalibration_amd import capi, host, pipeline, synth  # noqa: E402

print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
ctx = capi.Context(0)
grid = synth.make_grid(seed=12345, rows=10, cols=40, feats=64)
images, shape = pipeline.synthetic_views(ctx, grid, seed=7)
start = pipeline.perturbed_orientations(grid, 0.1, 99)
prev = None
for step in range(5):
    f0 = resource.getrusage(resource.RUSAGE_SELF).ru_minflt
    c0 = time.process_time()
    g, res, t = pipeline.run(ctx, grid, images, shape, start, relax=False)
    f1 = resource.getrusage(resource.RUSAGE_SELF).ru_minflt
    print(f"step {step}: minor faults {f1 - f0} ({(f1 - f0) / grid.n_images:.0f} per image), cpu {time.process_time() - c0:.2f} s", flush=True)
    if prev is not None:
        prev.close()
    prev = g
