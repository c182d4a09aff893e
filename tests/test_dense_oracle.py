"""The restated dense guided matching (oracle/dense.cpp) against the pieces of the reference that compile here
(oracle/_ref: hilbert.hpp, the jk::KDTree queries densifyMesh makes) and against the ground truth of a synthetic scene."""
import numpy as np
import pytest

from oracle import pyoracle
from dense_fixtures import dense_scene, ground_mesh_arrays


def _ref():
    from conftest import require_ref

    return require_ref(pyoracle, "ref_hilbert_xy2d")


def test_hilbert_index_is_the_reference_header():
    import ctypes as C
    r = _ref()
    r.ref_hilbert_xy2d.restype = C.c_uint32
    rng = np.random.default_rng(0)
    for order in (2, 8, 1024, 4096):
        for x, y in rng.integers(0, order, (200, 2)):
            assert pyoracle.hilbert_xy2d(order, x, y) == r.ref_hilbert_xy2d(int(order), int(x), int(y))
    # a bijection on the square (what makes it a walk order)
    assert sorted(pyoracle.hilbert_xy2d(8, x, y) for x in range(8) for y in range(8)) == list(range(64))


def test_reference_kdtree_queries_are_exhaustive_scans():
    """densifyMesh asks the reference's KD-tree for the 11 nearest cameras of a 3-D point and for the features in a disc
    (`squared distance < radius^2`); the restatement scans.  Same sets, same nearest-first order."""
    import ctypes as C
    r = _ref()
    f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
    u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
    r.ref_knn3.restype = C.c_size_t
    r.ref_knn3.argtypes = [f64p, C.c_size_t, f64p, C.c_size_t, u64p]
    r.ref_ball2.restype = C.c_size_t
    r.ref_ball2.argtypes = [f64p, C.c_size_t, f64p, C.c_double, u64p]
    rng = np.random.default_rng(3)
    cams = rng.uniform(0, 500, (300, 3)) * [1, 1, 0.02] + [0, 0, 100]
    out = np.zeros(300, np.uint64)
    for _ in range(50):
        q = rng.uniform(-50, 550, 3) * [1, 1, 0]
        k = r.ref_knn3(cams, len(cams), q, 11, out)
        d = np.sum((cams - q) ** 2, axis=1)
        assert list(out[:k]) == list(np.argsort(d, kind="stable")[:11])
    pts = rng.uniform(0, 1000, (2000, 2))
    pts[7] = [400.0, 300.0 + 150.0]                      # exactly on the circle: outside (`<`)
    got = np.zeros(2000, np.uint64)
    for q in [np.array([400.0, 300.0])] + [rng.uniform(0, 1000, 2) for _ in range(30)]:
        k = r.ref_ball2(pts, len(pts), q, 150.0 ** 2, got)
        d = np.sum((pts - q) ** 2, axis=1)
        assert sorted(got[:k]) == sorted(np.nonzero(d < 150.0 ** 2)[0])
        assert list(d[got[:k].astype(int)]) == sorted(d[got[:k].astype(int)])      # nearest first
    k = r.ref_ball2(pts, len(pts), np.array([400.0, 300.0]), 150.0 ** 2, got)
    assert 7 not in got[:k]


def test_restated_densify_recovers_the_ground():
    scene = dense_scene()
    v, e = ground_mesh_arrays(pyoracle.rebuild_mesh, scene)
    surface = pyoracle.RxSurface().set(v, e)
    n = len(scene["features"])
    out = pyoracle.densify_mesh(scene["position"], scene["orientation"], np.tile(scene["model"], (n, 1)), scene["features"],
                                scene["num_sparse"], surface)
    assert out["matches"] > 2000 and out["tracks"] > 500 and len(out["points"]) > 0.9 * out["tracks"]
    # measurement ids -> ground point ids: accepted matches join observations of the same ground point
    offs = np.concatenate([[0], np.cumsum([len(f[0]) - int(s) for f, s in zip(scene["features"], scene["num_sparse"])])])
    point_of = np.concatenate(scene["point_of"])
    a, b = point_of[out["match_pairs"][:, 0].astype(int)], point_of[out["match_pairs"][:, 1].astype(int)]
    assert np.mean((a == b) & (a >= 0)) > 0.995
    assert offs[-1] == len(point_of)
    # and the triangulated points lie on the ground they were generated from
    dz = out["points"][:, 2] - scene["ground"](out["points"][:, 0], out["points"][:, 1])
    assert np.median(np.abs(dz)) < 0.15 and np.mean(np.abs(dz) < 1.0) > 0.97
    assert len(surface.arrays()["cloud"]) == len(out["points"])


def test_host_hilbert_index_by_table_is_the_restatement():
    """liboc_host's index takes four levels per table look-up (a four-state machine over the original bits); the
    restatement walks level by level like types/hilbert.hpp.  Exhaustive on the small squares, sampled on the large ones,
    and the arguments the table does not take (order not a power of two, points outside the square) go level by level."""
    from opencalibration_amd import host
    L = host.load()
    for levels in range(1, 7):
        order = 1 << levels
        for x in range(order):
            for y in range(order):
                assert L.och_hilbert_xy2d(order, x, y) == pyoracle.hilbert_xy2d(order, x, y), (order, x, y)
    rng = np.random.default_rng(5)
    for levels in range(7, 17):
        order = 1 << levels
        for x, y in rng.integers(0, order, (300, 2)):
            assert L.och_hilbert_xy2d(order, int(x), int(y)) == pyoracle.hilbert_xy2d(order, int(x), int(y)), (order, x, y)
    for order, x, y in ((8192, 5471, 3647), (8192, 0, 0), (8192, 8191, 8191), (6, 3, 5), (1000, 999, 17), (16, 20, 3), (16, -1, 2), (1, 0, 0)):
        assert L.och_hilbert_xy2d(order, x, y) == pyoracle.hilbert_xy2d(order, x, y), (order, x, y)
