"""The suppression rules of AKAZE's scale-space maxima in numbers, on the same candidates of rendered views: OpenCV 4.x's three mask
passes (what oracle/akaze.cpp and the device run since round 6), the symmetric order-free rule of rounds 2 - 5, and the 3.x
running list as recalled; and the 4.x passes evaluated in dependency rounds (the device's schedule) against their sequential
form - which must agree on every candidate.  CPU only (the restatement).  usage: akaze_d1_census.py [n_views=4]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opencalibration_amd import synth  # noqa: E402  (data generators only)
from oracle import pyoracle  # noqa: E402

n_views = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = pyoracle.lib()
L.oc_akaze_suppression_census.argtypes = [np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS"), C.c_int, C.c_int,
                                          np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")]
tot = np.zeros(11, np.uint64)
for k in range(n_views):
    img = synth.render_blobs(1600, 1200, 100 + k)
    rng = np.random.default_rng(k)
    gray = np.clip(img[:, :, 0].astype(np.int32) + rng.integers(0, 20, (1200, 1600)) - 10, 0, 255).astype(np.uint8)
    c = np.zeros(11, np.uint64)
    L.oc_akaze_suppression_census(np.ascontiguousarray(gray), 1600, 1200, c)
    tot += c
    print("view %d: %d candidates; survivors: order-free %d, 3.x list %d, 4.x masks %d; candidates decided differently: "
          "order-free vs 3.x %d, order-free vs 4.x %d, 3.x vs 4.x %d; the 4.x masks in rounds differ from the sequential form on %d "
          "candidates, rounds per pass %d / %d / %d" % (k, *c), flush=True)
print("all %d views: %d candidates; survivors %d / %d / %d; decided differently %d (%.2f %% of the order-free survivors) / %d (%.2f %%) / %d"
      % (n_views, tot[0], tot[1], tot[2], tot[3], tot[4], 100.0 * tot[4] / tot[1], tot[5], 100.0 * tot[5] / tot[1], tot[6]))
assert tot[7] == 0
