"""ctypes binding of liboc_host.so (include/oc_host.h): the C++ host side of the hot path."""
import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboc_host.so")

_lib = None
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise capi.OchipError(f"{LIB_PATH} is missing: run `python -m opencalibration_amd.build`")
        capi.load()  # libochip.so first (liboc_host.so links against it)
        L = C.CDLL(LIB_PATH)
        L.och_subsample.restype = C.c_size_t
        L.och_subsample.argtypes = [_f64p, _f32p, C.c_size_t, C.c_double, C.c_size_t, _u64p]
        L.och_matches_from_device.restype = C.c_size_t
        L.och_matches_from_device.argtypes = [C.c_void_p, _u64p, C.c_size_t, _u64p, C.c_size_t, _u64p, _u64p, _f64p]
        _lib = L
    return _lib


def subsample(loc, strength, spacing, count=0):
    loc = np.ascontiguousarray(loc, np.float64)
    strength = np.ascontiguousarray(strength, np.float32)
    out = np.zeros(max(len(strength), 1), np.uint64)
    n = load().och_subsample(loc, strength, len(strength), spacing, count, out)
    return out[:n].copy()


def matches_from_device(raw, idx1, idx2):
    """raw: MATCH_DTYPE rows of one pair (len == len(idx1))."""
    raw = np.ascontiguousarray(raw, capi.MATCH_DTYPE)
    idx1 = np.ascontiguousarray(idx1, np.uint64)
    idx2 = np.ascontiguousarray(idx2, np.uint64)
    n = max(len(idx1), 1)
    i1, i2, d = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.float64)
    if len(raw) < len(idx1):
        raise ValueError("raw shorter than idx1")
    m = load().och_matches_from_device(raw.ctypes.data, idx1, len(idx1), idx2, len(idx2), i1, i2, d)
    return i1[:m].copy(), i2[:m].copy(), d[:m].copy()
