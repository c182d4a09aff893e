"""Mesh refinement of the host library (refine_mesh.cpp) against the reference's own test cases
(test/test_refine_mesh.cpp: 881-1033 counts / density refinement / variance filter, 1206-1310 locator vs brute force,
the mesh validity helpers :17-195 restated) and against the real ankerl::unordered_dense for the erase order the
refinement's choices depend on (oracle/_ref).  No device involved."""
import numpy as np
import pytest

from opencalibration_amd import host
from oracle import pyoracle

NONE = np.uint64(0xFFFFFFFFFFFFFFFF)


def _segments_cross(p1, p2, p3, p4):
    for a in (p1, p2):
        for b in (p3, p4):
            if np.sum((a - b) ** 2) < 1e-10:
                return False
    cr = lambda o, a, b: (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    d1, d2, d3, d4 = cr(p3, p4, p1), cr(p3, p4, p2), cr(p1, p2, p3), cr(p1, p2, p4)
    return ((d1 > 0 > d2) or (d1 < 0 < d2)) and ((d3 > 0 > d4) or (d3 < 0 < d4))


def validate_mesh(surface):
    """validateMeshNoCrossingEdges + validateMeshNoHangingNodes + adjacency consistency; returns the triangle set."""
    a = surface.arrays()
    v, e = a["vertices"][:, :2], a["edges"]
    segs = [(v[int(s)], v[int(d)]) for s, d in e[:, :2]]
    for i in range(len(segs)):
        for j in range(i + 1, len(segs)):
            assert not _segments_cross(*segs[i], *segs[j]), (i, j)
    for n, p in enumerate(v):
        for (s, d), (p1, p2) in zip(e[:, :2], segs):
            if n in (int(s), int(d)):
                continue
            ab, ap = p2 - p1, p - p1
            if abs(ab[0] * ap[1] - ab[1] * ap[0]) > 1e-9 * np.linalg.norm(ab):
                continue
            t = ap @ ab
            assert not (1e-9 < t < ab @ ab - 1e-9), "hanging node %d" % n
    tris = {}
    for s, d, border, o0, o1 in e:
        for o in ([o0] if border else [o0, o1]):
            assert o != NONE
            tris.setdefault(tuple(sorted((int(s), int(d), int(o)))), 0)
            tris[tuple(sorted((int(s), int(d), int(o))))] += 1
    assert all(c == 3 for c in tris.values()), "a triangle must be named by each of its three edges"
    return set(tris)


def _square(z=(0, 0, 0, 0)):
    """test_refine_mesh.cpp:946-965: the unit square of two triangles, 10 m wide."""
    v = np.array([[0, 0, z[0]], [10, 0, z[1]], [0, 10, z[2]], [10, 10, z[3]]], float)
    e = np.array([[0, 1, 1, 3, NONE], [0, 2, 1, 3, NONE], [1, 3, 1, 0, NONE], [2, 3, 1, 0, NONE], [0, 3, 0, 1, 2]], np.uint64)
    return host.Surface().set(v, e)


def _grid_cloud(lo, hi, step, zf):
    xs = np.arange(lo, hi, step)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), zf(X.ravel(), Y.ravel())], -1)


def test_unordered_dense_erase_moves_the_last_element():
    """The restated containers (vectors with swap-and-pop) are the real header's behaviour."""
    import ctypes as C
    from conftest import require_ref

    r = require_ref(pyoracle, "ref_dense_set_order")
    r.ref_dense_set_order.restype = C.c_size_t
    r.ref_dense_set_order.argtypes = [np.ctypeslib.ndpointer(np.int64), C.c_size_t, np.ctypeslib.ndpointer(np.uint64)]
    rng = np.random.default_rng(1)
    for _ in range(20):
        ops, model, live = [], [], set()
        for step in range(200):
            if live and rng.uniform() < 0.35:
                k = int(rng.choice(sorted(live)))
                ops.append(-k - 1)
                i = model.index(k)
                model[i] = model[-1]
                model.pop()
                live.discard(k)
            else:
                k = int(rng.integers(0, 1 << 40))
                if k in live:
                    continue
                ops.append(k)
                model.append(k)
                live.add(k)
        out = np.zeros(len(ops) + 1, np.uint64)
        n = r.ref_dense_set_order(np.array(ops, np.int64), len(ops), out)
        assert list(out[:n]) == model


def test_count_points_per_triangle():
    s = host.rebuild_mesh(np.array([[0, 0, 10], [5, 5, 10.0]]))
    s.set_clouds([[[2.5, 2.5, 0], [1, 1, 0], [4, 4, 0]]])
    tri, count, var = s.count_points_per_triangle()
    assert count.sum() == 3
    # variance_filters_coplanar_points (:946-988)
    sq = _square()
    sq.set_clouds([_grid_cloud(0.5, 10, 0.3, lambda x, y: 0 * x)])
    tri, count, var = sq.count_points_per_triangle()
    assert len(tri) == 2 and (count > 20).all() and np.abs(var).max() < 1e-10
    assert sq.refine_by_point_density(20, 0.01, 10) == 0 and len(sq.arrays()["vertices"]) == 4
    # variance_triggers_refinement_for_uneven_surface (:990-1033)
    sq = _square()
    sq.set_clouds([_grid_cloud(0.5, 10, 0.3, lambda x, y: np.sin(x) * 0.5)])
    tri, count, var = sq.count_points_per_triangle()
    assert (count > 20).all() and (var > 0.01).all()
    assert sq.refine_by_point_density(20, 0.01, 10) > 0 and len(sq.arrays()["vertices"]) > 4
    validate_mesh(sq)


def test_count_is_in_first_point_order_with_exact_sums():
    sq = _square(z=(0, 1, 2, 3))
    pts = _grid_cloud(0.25, 10, 0.5, lambda x, y: np.sin(x) * np.cos(y))
    sq.set_clouds([pts[:150], pts[150:]])
    tri, count, var = sq.count_points_per_triangle()
    where = sq.locate(pts[:, :2])
    keys = [tuple(sorted(int(x) for x in t)) for t in where]
    first = []
    for k in keys:
        if k not in first:
            first.append(k)
    assert [tuple(sorted(int(x) for x in t)) for t in tri] == first
    v = sq.arrays()["vertices"]
    for t, c, vv in zip(tri, count, var):
        k = tuple(sorted(int(x) for x in t))
        sel = pts[[kk == k for kk in keys]]
        p0, p1, p2 = v[int(t[0])], v[int(t[1])], v[int(t[2])]
        n = np.cross(p1 - p0, p2 - p0)
        n /= np.linalg.norm(n)
        d = (sel - p0) @ n
        assert c == len(sel) and abs(vv - (np.mean(d * d) - np.mean(d) ** 2)) < 1e-12


def test_refine_by_point_density_keeps_the_mesh_conforming():
    """refine_by_point_density (:906-944) on the minimal mesh, then deeper with a size limit."""
    s = host.rebuild_mesh(np.array([[0, 0, 10], [10, 10, 10.0]]), minimal=True)
    assert len(s.arrays()["vertices"]) == 4
    pts = _grid_cloud(1, 10, 0.5, lambda x, y: np.sin(x) * np.cos(y) * 2.0)
    s.set_clouds([pts])
    created = s.refine_by_point_density(20, 0.0, 10)
    a = s.arrays()
    tris = validate_mesh(s)
    assert created > 0 and len(a["vertices"]) > 4
    assert len(tris) == 2 + created // 2          # every bisection turns 1 triangle into 2 (2 or 4 created per edge)
    # Euler: V - E + F = 1 for a triangulated disc
    assert len(a["vertices"]) - len(a["edges"]) + len(tris) == 1
    # the densest triangles were split: no triangle keeps more than 20 points unless its plane fits them
    tri, count, var = s.count_points_per_triangle()
    assert count.max() <= 20 or var[count > 20].max() <= 0.0 + 1e-12
    # with a minimum triangle size nothing below it is split
    s2 = host.rebuild_mesh(np.array([[0, 0, 10], [10, 10, 10.0]]), minimal=True)
    s2.set_clouds([pts])
    s2.refine_by_point_density(20, 0.0, 10, min_triangle_size=6.0)
    v2, e2 = s2.arrays()["vertices"], s2.arrays()["edges"]
    longest = {}
    for t in validate_mesh(s2):
        p = v2[list(t), :2]
        longest[t] = max(np.linalg.norm(p[0] - p[1]), np.linalg.norm(p[1] - p[2]), np.linalg.norm(p[2] - p[0]))
    assert min(longest.values()) >= 6.0 / 2 - 1e-9   # a split halves at most one generation below the limit


def test_locator_matches_brute_force():
    """triangle_locator_matches_brute_force_* (:1206-1310): minimal, grid and refined meshes, points inside and outside."""
    cams = np.array([[x, y, 50.0] for x in range(0, 100, 20) for y in range(0, 80, 20)], float)
    meshes = [host.rebuild_mesh(np.array([[0, 0, 10], [10, 10, 10.0]]), minimal=True), host.rebuild_mesh(cams)]
    refined = host.rebuild_mesh(cams)
    for x, y in [(30, 30), (31, 29), (60, 10), (5, 70)]:
        refined.refine_at_point(x, y, 3)
    validate_mesh(refined)
    meshes.append(refined)
    for s in meshes:
        a = s.arrays()
        v = a["vertices"]
        tris = validate_mesh(s)
        lo, hi = v[:, :2].min(0) - 7, v[:, :2].max(0) + 7
        q = np.stack(np.meshgrid(np.linspace(lo[0], hi[0], 37), np.linspace(lo[1], hi[1], 31), indexing="ij"), -1).reshape(-1, 2)
        got = s.locate(q)
        for p, t in zip(q, got):
            inside = []
            for tri in tris:
                a0, a1, a2 = v[list(tri), :2]
                d = [(p[0] - b[0]) * (c[1] - b[1]) - (c[0] - b[0]) * (p[1] - b[1]) for b, c in ((a1, a0), (a2, a1), (a0, a2))]
                if not (min(d) < 0 and max(d) > 0):
                    inside.append(tri)
            if t[0] == NONE:
                assert not inside, p
            else:
                assert tuple(sorted(int(x) for x in t)) in inside, p


def test_refine_at_point_levels_and_depth():
    """refine_at_point_single_level / multiple_levels (:325-368)."""
    s = host.rebuild_mesh(np.array([[0, 0, 10], [10, 10, 10.0]]), minimal=True)
    n0 = len(validate_mesh(s))
    c1 = s.refine_at_point(2.0, 3.0, 1)
    assert c1 in (2, 4) and len(validate_mesh(s)) == n0 + c1 // 2
    c3 = s.refine_at_point(2.0, 3.0, 3)
    assert c3 >= 6
    validate_mesh(s)
    assert s.refine_at_point(1e6, 1e6, 2) == 0            # outside: nothing to refine


# ---- the host refinement against the second restatement (oracle/refine_mesh.cpp, written from the reference's text) ----
def _oracle_twin(surface):
    a = surface.arrays()
    return pyoracle.RxMesh(a["vertices"], a["edges"])


def _same_mesh(surface, rx):
    a = surface.arrays()
    v, e = rx.arrays()
    return np.array_equal(a["vertices"], v) and np.array_equal(a["edges"], e)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_refinement_equals_the_oracle_restatement(seed):
    """countPointsPerTriangle (rows, order, counts, variances) and several rounds of refineByPointDensity with the heights
    moved between the rounds (as the relax stage moves them): the two restatements keep identical meshes - vertices in
    creation order, edges in the container's iteration order - and identical counts of created triangles."""
    rng = np.random.default_rng(seed)
    cams = np.array([[x, y, 50.0] for x in np.arange(0, 60 + 10 * seed, 20) for y in np.arange(0, 60, 20)], float)
    cams[:, :2] += rng.uniform(-2, 2, (len(cams), 2))
    s = host.rebuild_mesh(cams, minimal=(seed % 2 == 0))
    rx = _oracle_twin(s)
    assert _same_mesh(s, rx)
    lo, hi = cams[:, :2].min(0) - 5, cams[:, :2].max(0) + 5
    total = 0
    for rnd in range(4):
        pts = np.concatenate([rng.uniform(lo, hi, (900, 2)), np.zeros((900, 1))], axis=1)
        pts[:, 2] = 1.5 * np.sin(pts[:, 0] / 7.0 + rnd) * np.cos(pts[:, 1] / 5.0) + rng.normal(0, 0.02, len(pts))
        clouds = [pts[:400], pts[400:]]
        s.set_clouds(clouds)
        ht, hc, hv = s.count_points_per_triangle()
        ot, oc, ov = rx.count_points_per_triangle(clouds)
        assert np.array_equal(ht, ot) and np.array_equal(hc, oc) and np.array_equal(hv, ov)
        created_h = s.refine_by_point_density(20, 0.01, 2, min_triangle_size=1.0)
        created_o = rx.refine_by_point_density(clouds, 20, 0.01, 2, min_triangle_size=1.0)
        assert created_h == created_o and _same_mesh(s, rx), (rnd, created_h, created_o)
        total += created_h
        # the relax moves the heights and hands the mesh back with its containers untouched
        a = s.arrays()
        z = a["vertices"][:, 2] + rng.normal(0, 0.05, len(a["vertices"]))
        s.set_heights(z)
        rx.set_heights(z)
    assert total > 10
    validate_mesh(s)


def test_refine_at_point_equals_the_oracle_restatement():
    cams = np.array([[x, y, 50.0] for x in range(0, 100, 20) for y in range(0, 80, 20)], float)
    s = host.rebuild_mesh(cams)
    rx = _oracle_twin(s)
    for x, y, lv in [(30, 30, 3), (31, 29, 2), (60, 10, 4), (5, 70, 1), (1e6, 0, 2), (47.3, 33.1, 5)]:
        assert s.refine_at_point(x, y, lv) == rx.refine_at_point(x, y, lv)
        assert _same_mesh(s, rx), (x, y)
