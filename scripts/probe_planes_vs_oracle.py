#!/usr/bin/env python3
"""The device's Lt / (Lx, Ly) pyramids of one view (OCHIP_DUMP_PLANES) against the oracle's (oc_akaze_level), level by
level: where do they first differ?  usage: probe_planes_vs_oracle.py [w=640] [h=480] [noise amplitude=40]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from opencalibration_amd import capi, synth
from oracle import pyoracle

w = int(sys.argv[1]) if len(sys.argv) > 1 else 640
h = int(sys.argv[2]) if len(sys.argv) > 2 else 480
amp = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rng = np.random.default_rng(w)
base = synth.render_blobs(w, h, 31)
img = base if amp == 0 else np.clip(base.astype(np.int32) + rng.integers(0, amp, (h, w, 1)) - amp // 2, 0, 255).astype(np.uint8)
os.environ["OCHIP_DUMP_PLANES"] = "/tmp/pvo"
ctx = capi.Context(0)
got, wh = ctx.akaze_batch(img[None], max_kp=60000)
lt = np.fromfile("/tmp/pvo_lt.f32", np.float32)
lxy = np.fromfile("/tmp/pvo_lxy.f32", np.float32)
L = pyoracle.lib()
L.oc_akaze_level.restype = C.c_size_t
gray = np.ascontiguousarray(img[:, :, 0])
kc = np.zeros(1, np.float32)
kp = np.zeros((60000, 6), np.float32)
d = np.zeros((60000, 8), np.uint64)
L.oc_akaze.argtypes = None
n = L.oc_akaze(gray.ctypes.data_as(C.c_void_p), C.c_int(w), C.c_int(h), C.c_size_t(60000), kp.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
               kc.ctypes.data_as(C.c_void_p))
print("oracle keypoints", n, "kcontrast", kc[0], "device keypoints", len(got[0][0]))
off = 0
for lvl in range(16):
    lw, lh = w >> (lvl // 4), h >> (lvl // 4)
    if lvl // 4 and (lw < 80 or lh < 40):
        break
    out = np.zeros(lw * lh, np.float32)
    a, b = C.c_int(0), C.c_int(0)
    res = []
    for which, name in ((0, "Lt"), (1, "Lx"), (2, "Ly")):
        L.oc_akaze_level.argtypes = None
        L.oc_akaze_level(gray.ctypes.data_as(C.c_void_p), C.c_int(w), C.c_int(h), C.c_int(lvl), C.c_int(which), out.ctypes.data_as(C.c_void_p), C.byref(a), C.byref(b))
        o = out.reshape(lh, lw)
        if which == 0:
            g = lt[off:off + lw * lh].reshape(lh, lw)
        else:
            g = lxy[2 * off:2 * (off + lw * lh)].reshape(lh, lw, 2)[:, :, which - 1]
        bad = np.argwhere(g.view(np.uint32) != o.view(np.uint32))
        res.append("%s %d differ%s" % (name, len(bad), "" if not len(bad) else " (max |d| %.3g, rows %d..%d cols %d..%d)" % (
            np.abs(g - o).max(), bad[:, 0].min(), bad[:, 0].max(), bad[:, 1].min(), bad[:, 1].max())))
    print("level %2d %4d x %4d: %s" % (lvl, lw, lh, "; ".join(res)))
    off += lw * lh
