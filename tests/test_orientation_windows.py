"""The orientation kernels decide "which of the 42 sliding windows contain this angle" from a table of the windows'
84 ends instead of evaluating the restatement's comparison chain per window (oracle/akaze.cpp:606-614, following
AKAZE's Compute_Main_Orientation).  Host-only check (no device): the table and the chain agree on every float tried -
random bit patterns over (0, 2 pi), a dense sweep, every window end with its float neighbours, and the values outside
the open interval."""
import ctypes as C

import numpy as np

from opencalibration_amd import capi


def _both(angles):
    L = capi.load()
    a = np.ascontiguousarray(angles, np.float32)
    t, p = np.zeros(len(a), np.uint64), np.zeros(len(a), np.uint64)
    L.ochip_debug_orientation_windows.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    assert L.ochip_debug_orientation_windows(a.ctypes.data, len(a), t.ctypes.data, p.ctypes.data) == 0
    return t, p


def test_table_equals_predicate():
    two_pi = np.float32(6.28318530717958647692)
    rng = np.random.default_rng(7)
    # random bit patterns of positive floats below 2 pi (every exponent, not a uniform sweep)
    bits = rng.integers(0, np.float32(two_pi).view(np.uint32), 2_000_000, dtype=np.uint32)
    sweep = np.linspace(0, float(two_pi), 1_000_001, dtype=np.float32)
    # the window ends: starts accumulate 0.15f in float, ends are start + pi/3 or start - 5 pi/3
    starts, a = [], np.float32(0)
    while a < two_pi:
        starts.append(a)
        a = np.float32(a + np.float32(0.15))
    assert len(starts) == 42
    pi = np.float32(3.14159265358979323846)
    ends = [np.float32(s - np.float32(5) * pi / np.float32(3)) if np.float32(s + pi / np.float32(3)) > two_pi
            else np.float32(s + pi / np.float32(3)) for s in starts]
    near = []
    for e in starts + ends:
        x = np.float32(e)
        for _ in range(4):
            x = np.nextafter(x, np.float32(-np.inf))
        for _ in range(9):
            near.append(x)
            x = np.nextafter(x, np.float32(np.inf))
    special = np.array([0.0, -0.0, two_pi, np.nextafter(two_pi, np.float32(0)), 7.0, -1.0, np.inf, np.nan, 1e-38, 1e-45], np.float32)
    angles = np.concatenate([bits.view(np.float32), sweep, np.array(near, np.float32), special])
    t, p = _both(angles)
    assert np.array_equal(t, p)
    inside = (angles > 0) & (angles < two_pi)
    assert not t[~inside].any()
    # pi / 3 = 6.98 steps of 0.15, and the last step before 2 pi is shorter: 6 to 8 windows contain an angle
    pop = np.array([bin(int(v)).count("1") for v in t[:200_000]])
    assert pop.min() >= 6 and pop.max() <= 8 and (t < (1 << 42)).all()
