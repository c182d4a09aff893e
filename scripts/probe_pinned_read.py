import time, numpy as np, sys
sys.path.insert(0, "/root/repo")
from opencalibration_amd import capi
ctx = capi.Context(0)
a, rel = ctx.host_array((64, 1 << 20, 4))     # 256 MB page-locked, uint8
a[:] = 1
dst = np.empty_like(a)
dst[:] = 0                                     # touched
for name, src in (("pinned", a), ("pageable", dst.copy())):
    t0 = time.perf_counter(); np.copyto(dst, src); t = time.perf_counter() - t0
    print(name, "-> touched heap:", round(a.nbytes / t / 1e9, 2), "GB/s")
t0 = time.perf_counter(); fresh = np.empty_like(a); np.copyto(fresh, a); t = time.perf_counter() - t0
print("pinned -> fresh pages:", round(a.nbytes / t / 1e9, 2), "GB/s")
rel()
