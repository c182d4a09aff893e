"""Pins restated building blocks against the reference's OWN code: the std-only headers of /root/reference that compile in
this image (oracle/_ref/libref.so, built in place by oracle/Makefile `ref`): include/opencalibration/relax/grid_filter.hpp
(+ external/unordered_dense), types/union_find.hpp.  These are the pieces of a14 (grid filter of the inliers) and a15
(union-find tracks, per-cell longest-track filter) that decide WHICH residual blocks a relax problem gets."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    from conftest import require_ref

    r = require_ref(oracle, "ref_grid_filter_values")
    u64p, f64p = oracle.u64p, oracle.f64p
    for lib in (r, oracle.lib()):
        pre = "ref" if lib is r else "ocx"
        getattr(lib, pre + "_grid_filter_values").argtypes = [f64p, f64p, u64p, C.c_size_t, C.c_double, u64p, C.c_void_p]
        getattr(lib, pre + "_union_find").argtypes = [C.c_size_t, u64p, C.c_size_t, u64p]
        getattr(lib, pre + "_grid_cell_key").restype = C.c_uint64
        getattr(lib, pre + "_grid_cell_key").argtypes = [C.c_int, C.c_int]
    return r


def _filter(lib, pre, xy, score, value, res):
    out, n = np.zeros(len(value) + 1, np.uint64), C.c_size_t(0)
    getattr(lib, pre + "_grid_filter_values")(xy, score, value, len(value), res, out, C.addressof(n))
    return set(out[:n.value].tolist())


@pytest.mark.parametrize("res", [0.15, 0.1, 0.05, 0.075])
def test_grid_filter_matches_reference_header(oracle, ref, res):
    rng = np.random.default_rng(int(res * 1000))
    for trial in range(20):
        n = int(rng.integers(1, 800))
        xy = np.ascontiguousarray(rng.uniform(-0.05, 1.05, (n, 2)))  # includes cells left of / above the image
        score = np.ascontiguousarray(np.round(rng.uniform(0, 1, n), 2 if trial % 2 else 12))  # with and without score ties
        value = np.ascontiguousarray(np.arange(n, dtype=np.uint64))
        assert _filter(ref, "ref", xy, score, value, res) == _filter(oracle.lib(), "ocx", xy, score, value, res)
        # values that repeat (a track root entered once per image cell): the set semantics of _best
        value = np.ascontiguousarray(rng.integers(0, max(2, n // 3), n).astype(np.uint64))
        assert _filter(ref, "ref", xy, score, value, res) == _filter(oracle.lib(), "ocx", xy, score, value, res)


def test_grid_cell_key_matches_reference_header(oracle, ref):
    for i, j in [(0, 0), (3, 9), (-1, 2), (2, -1), (-1, -1), (2 ** 20, 5)]:
        assert ref.ref_grid_cell_key(i, j) == oracle.lib().ocx_grid_cell_key(i, j)


def test_union_find_matches_reference_header(oracle, ref):
    rng = np.random.default_rng(4)
    for trial in range(30):
        n = int(rng.integers(2, 2000))
        pairs = np.ascontiguousarray(rng.integers(0, n, (int(rng.integers(1, 2 * n)), 2)).astype(np.uint64))
        a, b = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        ref.ref_union_find(n, pairs, len(pairs), a)
        oracle.lib().ocx_union_find(n, pairs, len(pairs), b)
        assert np.array_equal(a, b)  # the same ROOT ids, not just the same partition: track order follows the roots


def test_kmeans_matches_reference_header(oracle, ref):
    """include/opencalibration/geometry/KMeans.hpp compiled in place against the restatement used by the partitioning
    of the relax stage: same clusters (sizes, centroids bit for bit, assignment), after 0..10 iterate() calls."""
    rng = np.random.default_rng(11)
    for trial in range(12):
        n, k = int(rng.integers(20, 600)), int(rng.integers(2, 9))
        pts = rng.normal(size=(n, 3)) * rng.uniform(0.5, 20, 3)
        if trial % 3 == 0:   # clumps: small clusters get re-seeded next to big ones (KMeans.hpp:196-214)
            pts[: n // 2] = pts[: n // 2] * 0.01 + 50
        for iters in (1, 2, 5, 11):
            a1, c1, s1 = oracle.kmeans3(pts, k, iters, use_ref=True)
            a2, c2, s2 = oracle.kmeans3(pts, k, iters, use_ref=False)
            assert np.array_equal(s1, s2) and np.array_equal(c1, c2) and np.array_equal(a1, a2), (trial, iters)
