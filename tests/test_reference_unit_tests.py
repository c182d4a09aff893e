"""The reference's own unit tests for the geometry primitives, the plain cost functors and the surface mesh, restated case by
case with their values and tolerances (test/test_geometry.cpp:10-230, test/test_cost_functions.cpp:7-106,
test/test_meshgraph.cpp:14-174) and run against the restatement under oracle/ - and, for the mesh construction, against the host
library too.  These are known-answer pins of the oracle: every number below is the reference's.  No device."""
import ctypes as C

import numpy as np
import pytest

from opencalibration_amd import host

D3 = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
INTERSECTION, OUTSIDE_BORDER, GRAPH_STRUCTURE_INCONSISTENT = 2, 3, 5   # MeshIntersectionSearcher::IntersectionInfo (intersect.hpp)


@pytest.fixture(scope="module")
def rx(oracle):
    L = oracle._rx()
    L.ocx_ray_intersection.argtypes = [D3, D3, D3, D3, D3]
    L.ocx_corner_plane.argtypes = [D3, D3, D3]
    L.ocx_ray_plane.argtypes = [D3, D3, D3, D3, D3]
    L.ocx_ray_plane.restype = C.c_int
    L.ocx_angle_between_unit_vectors.argtypes = [D3, D3]
    L.ocx_angle_between_unit_vectors.restype = C.c_double
    L.ocx_difference_cost.argtypes = [C.c_double] * 3
    L.ocx_difference_cost.restype = C.c_double
    L.ocx_distortion_monotonicity.argtypes = [C.c_double, C.c_double, D3, D3]
    L.ocx_adjacent_triangle_normal.argtypes = [D3, D3, C.c_double]
    L.ocx_adjacent_triangle_normal.restype = C.c_double
    L.ocx_surface_intersect.argtypes = [C.c_void_p, D3, D3, D3, np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")]
    L.ocx_surface_intersect.restype = C.c_int
    return L


def v(*x):
    return np.array(x, np.float64)


def ray_intersection(rx, d1, o1, d2, o2):
    out = np.zeros(4)
    rx.ocx_ray_intersection(v(*d1), v(*o1), v(*d2), v(*o2), out)
    return out[:3], out[3]


# ---- test_geometry.cpp (a ray_d is {dir, offset})
def test_ray_intersection_nan_infinite_intersection(rx):          # :10-21
    p, e = ray_intersection(rx, (0, 0, 1), (0, 0, 0), (0, 0, 1), (0, 0, 0))
    assert np.isnan(p).all() and np.isnan(e)


def test_ray_intersection_never(rx):                              # :23-34
    p, e = ray_intersection(rx, (0, 0, 1), (0, 0, 0), (0, 0, 1), (1, 0, 0))
    assert np.isnan(p).all() and np.isnan(e)


def test_ray_intersection_exact_at_origin(rx):                    # :36-48
    p, e = ray_intersection(rx, (0, 0, 1), (0, 0, -1), (0, 1, 0), (0, 10, 0))
    assert np.linalg.norm(p - v(0, 0, 0)) == 0 and e == 0          # (EXPECT_DOUBLE_EQ with 0: exact)


def test_ray_intersection_offset_from_origin(rx):                 # :50-62
    p, e = ray_intersection(rx, (0, 0, 1), (1, 0, 1), (0, 1, 0), (1, 1, 0))
    assert np.linalg.norm(p - v(1, 0, 0)) == 0 and e == 0


def test_ray_intersection_inexact(rx):                            # :64-85
    p, e = ray_intersection(rx, (0, 0, 1), (2, 0, 1), (0, 1, 0), (0, 1, 0))
    assert np.linalg.norm(p - v(1, 0, 0)) == 0 and e == -2 * 2     # behind the normals: negative
    p, e = ray_intersection(rx, (0, 0, -1), (2, 0, 1), (0, -1, 0), (0, 1, 0))
    assert np.linalg.norm(p - v(1, 0, 0)) == 0 and e == 2 * 2


def test_plane_conversion(rx):                                    # :87-102
    norm, off = np.zeros(3), np.zeros(3)
    rx.ocx_corner_plane(v(0, 0, -2, 1, 0, -2, 0, 1, -2), norm, off)
    assert np.linalg.norm(v(0, 0, -2) - off) < 1e-9 and np.linalg.norm(v(0, 0, 1) - norm) < 1e-9


def test_ray_plane_intersection(rx):                              # :104-121
    out = np.zeros(3)
    assert rx.ocx_ray_plane(v(0, 0, 0.5), v(3, 3, -10), v(0, 0, 1), v(5, 5, 2), out) == 1
    assert np.linalg.norm(v(3, 3, 2) - out) < 1e-9


def test_ray_parallel_to_the_plane(rx):
    """rayTriangleIntersection's parallel case (:145-164) as far as the restatement goes: the path's code calls
    rayPlaneIntersection and a walk over mesh triangles (below), never rayTriangleIntersection itself - its in / out edge cases
    (:123-230) belong to the orthomosaic, outside the path."""
    out = np.zeros(3)
    norm, off = np.zeros(3), np.zeros(3)
    rx.ocx_corner_plane(v(0, 0, -2, 10, 0, -2, 0, 10, -2), norm, off)
    assert rx.ocx_ray_plane(v(1, 0, 0), v(0.5, 0.5, -10), norm, off, out) == 0 and np.isnan(out).all()
    assert rx.ocx_ray_plane(v(0, 0, 0.5), v(0.5, 0.5, -10), norm, off, out) == 1
    assert np.linalg.norm(out - v(0.5, 0.5, -2)) < 1e-9           # (:123-143's expected point)


# ---- test_cost_functions.cpp
def test_difference_cost(rx):                                     # :7-23
    assert rx.ocx_difference_cost(2.0, 5.0, 3.0) == 4.0
    assert rx.ocx_difference_cost(1.0, 7.0, 7.0) == 0.0


def test_distortion_monotonicity(rx):                             # :25-54
    res = np.full(10, np.nan)
    rx.ocx_distortion_monotonicity(1.0, 1.0, v(0, 0, 0), res)
    assert np.array_equal(res, np.zeros(10))
    rx.ocx_distortion_monotonicity(1.0, 1.0, v(-10.0, 0, 0), res)
    assert (res >= 0).all() and (res > 0).any()


def test_adjacent_triangle_normal(rx):                            # :56-78
    assert abs(rx.ocx_adjacent_triangle_normal(v(0, 0, 1, 0, 0.5, 1, 0.5, 0.5), v(0, 0, 0, 0), 1.0)) <= 1e-5      # coplanar
    assert abs(rx.ocx_adjacent_triangle_normal(v(0, 0, 1, 0, 0.5, 1, 0.5, -1), v(0, 0, 0, 5.0), 1.0)) > 0.1      # not


def test_robust_centroid(oracle):                                 # :80-101
    c = oracle.robust_centroid(np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float64), 10.0)
    assert np.allclose(c, 1.0 / 3, atol=1e-6)
    c = oracle.robust_centroid(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [100, 100, 100]], np.float64), 1.0)
    assert np.linalg.norm(c - v(1.0 / 3, 1.0 / 3, 0)) < np.linalg.norm(c - v(100, 100, 100))


def test_angle_between_unit_vectors(rx):                          # :103-111
    assert abs(rx.ocx_angle_between_unit_vectors(v(1, 0, 0), v(0, 1, 0)) - np.pi / 2) < 1e-10
    assert abs(rx.ocx_angle_between_unit_vectors(v(1, 0, 0), v(1, 0, 0))) < 1e-5


# ---- test_meshgraph.cpp: rebuildMesh(point_cloud, {surface_model{{}, empty mesh}}) and the triangle walk
def _counts(s):
    a = s.arrays()
    return len(a["vertices"]), len(a["edges"])


@pytest.mark.parametrize("lib", ["oracle", "host"])
def test_mesh_expansion_counts(oracle, lib):                      # :14-45
    rebuild = oracle.rebuild_mesh if lib == "oracle" else host.rebuild_mesh
    assert _counts(rebuild(np.zeros((0, 3)))) == (0, 0)
    assert _counts(rebuild(v(0, 0, 0).reshape(1, 3))) == (0, 0)
    assert _counts(rebuild(np.array([[0, 0, 0], [1, 0, 0]], np.float64))) == (30, 69)


def _intersect(rx, surface, x, y, z=0.0):
    loc, tri = np.zeros(3), np.zeros(3, np.uint64)
    t = rx.ocx_surface_intersect(surface.h, v(0, 0, 1), v(x, y, z), loc, tri)
    return t, loc


def test_mesh_intersects_rays(rx, oracle):                        # :47-72
    g = oracle.rebuild_mesh(np.array([[0, 0, 0], [1, 0, 0]], np.float64))
    for i in range(50):
        for j in range(50):
            x, y = -2 + j * (5. / 50), -2 + i * (4. / 50)
            t, loc = _intersect(rx, g, x, y)
            assert t == INTERSECTION and np.linalg.norm(v(x, y, -1) - loc) < 1e-9, (x, y, t, loc)


def test_mesh_cycle_on_vertex_resolves(rx, oracle):               # :74-124
    pts = np.array([[x, y, 0.0] for x in np.arange(-2, 2.25, 0.5) for y in np.arange(-2, 2.25, 0.5)])
    g = oracle.rebuild_mesh(pts)
    a = g.arrays()
    assert len(a["vertices"]) > 10
    for loc in a["vertices"]:                                      # rays at every vertex: they land on shared edges
        t, hit = _intersect(rx, g, loc[0], loc[1], loc[2] + 5)
        assert t != GRAPH_STRUCTURE_INCONSISTENT
        if t == INTERSECTION:
            assert abs(hit[2] - loc[2]) < 0.01
    for e in a["edges"]:                                           # ... and at every edge's midpoint
        mid = (a["vertices"][int(e[0])] + a["vertices"][int(e[1])]) * 0.5
        assert _intersect(rx, g, mid[0], mid[1], mid[2] + 5)[0] != GRAPH_STRUCTURE_INCONSISTENT


def test_mesh_doesnt_intersect_outside(rx, oracle):               # :126-174
    g = oracle.rebuild_mesh(np.array([[0, 0, 0], [1, 0, 0]], np.float64))
    for i in range(50):
        y, x = -2 + i * (4. / 50), -2 + i * (5. / 50)
        assert _intersect(rx, g, -2.01, y)[0] == OUTSIDE_BORDER
        assert _intersect(rx, g, 3.01, y)[0] == OUTSIDE_BORDER
        assert _intersect(rx, g, x, -2.01)[0] == OUTSIDE_BORDER
        assert _intersect(rx, g, x, 2.01)[0] == OUTSIDE_BORDER


# ---- test_combinatorics.cpp / test_tree.cpp: the reference's own headers, compiled in place (oracle/_ref/libref.so)
def test_interleave_and_kdtree_known_answers(oracle):
    """interleave.hpp (pipeline.cpp:553 deals the three stages' runners with it) and jk/KDTree.h under their own unit tests'
    known answers (test_combinatorics.cpp:22-37: a0 b0 c0 a1 b1 b2 a2 b3 b4 a3 b5; test_tree.cpp:10-38: the two nearest of
    (6, 6) among (1, 2), (1, 3), (7, 7) are (7, 7) then (1, 3)) - through the shim that compiles those headers where they lie."""
    from conftest import require_ref

    r = require_ref(oracle, "ref_interleave3")
    u64p = oracle.u64p
    r.ref_interleave3.restype = C.c_size_t
    r.ref_interleave3.argtypes = [u64p, C.c_size_t, u64p, C.c_size_t, u64p, C.c_size_t, C.c_int, u64p]
    a, b, c = np.arange(0, 4, dtype=np.uint64), np.arange(10, 16, dtype=np.uint64), np.array([20], np.uint64)
    out = np.zeros(16, np.uint64)
    n = r.ref_interleave3(a, 4, b, 6, c, 1, 1, out)                  # (full_dispersal = true, the default the test uses)
    assert out[:n].tolist() == [0, 10, 20, 1, 11, 12, 2, 13, 14, 3, 15]
    assert r.ref_interleave3(a, 0, b, 0, c, 0, 1, out) == 0
    xy = np.array([[1, 2], [1, 3], [7, 7]], np.float64)
    q = np.array([[6.0, 6.0]])
    # ref_knn: for every point of xy its k nearest (itself first); the monster's location as a fourth point
    pts = np.vstack([xy, q])
    knn = np.zeros((4, 3), np.uint64)
    r.ref_knn(np.ascontiguousarray(pts), 4, 3, knn)
    assert knn[3].tolist() == [3, 2, 1]                           # itself, Melvin (7, 7), Harold (1, 3)


# ---- test_relax.cpp:250-296: DecomposedRotationCost's residuals at the truth and at fixed offsets (fixture :30-37)
def _q(axis, angle):                                              # Eigen::AngleAxisd as a quaternion, x y z w
    a = np.asarray(axis, np.float64)
    return np.concatenate([np.sin(angle / 2) * a / np.linalg.norm(a), [np.cos(angle / 2)]])


def _qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return v(aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw,
             aw * bw - ax * bx - ay * by - az * bz)


def _qrot(q, x):
    qv = q[:3]
    uv = 2 * np.cross(qv, x)
    return x + q[3] * uv + np.cross(qv, uv)


def test_rel_rot_cost_function(oracle):
    L = oracle._rx()
    L.ocx_decomposed_rotation_cost.argtypes = [D3, D3, D3, D3, C.c_int, D3, D3, D3]
    down = _q((1, 0, 0), np.pi)
    ori = [_qmul(_q((0, 0, 1), 0.2), down), _qmul(_q((0, 1, 0), -0.3), down)]
    pos = [v(9, 9, 9), v(11, 9, 9)]
    inv0 = ori[0] * v(-1, -1, -1, 1)
    rel_rot, rel_pos = _qmul(ori[1], inv0), _qrot(inv0, pos[1] - pos[0])

    def cost(q0, q1):
        r = np.full(3, np.nan)
        L.ocx_decomposed_rotation_cost(rel_rot, rel_pos, pos[0], pos[1], 8, np.ascontiguousarray(q0), np.ascontiguousarray(q1), r)
        return r

    r = cost(ori[0], ori[1])                                       # a perfect guess
    assert abs(r[0]) < 1e-5 and abs(r[1]) < 1e-5 and abs(r[2]) < 1e-12
    r = cost(_qmul(ori[0], _q((0, 0, 1), 0.3)), ori[1])            # camera 0 turned by 0.3 about its axis
    assert abs(r[0] - 0.3) < 1e-12 and abs(r[1]) < 1e-5 and abs(r[2] - 0.3) < 1e-12
    r = cost(ori[0], _qmul(ori[1], _q((0, 0, 1), -0.3)))           # camera 1 turned by -0.3
    assert abs(r[0]) < 1e-5 and abs(r[1] - 0.3) < 1e-12 and abs(r[2] - 0.3) < 1e-12
