"""Pins oracle/relax_full.cpp (the whole relax stage restated: ground plane, ground mesh, RelaxGroup, meshes) with the
reference's own tests restated - test/test_relax.cpp:416-434 (plane), :833-1050 (incremental_relax, 5 x 5 grid through
RelaxGroup with depth-2/3 context) - and against the first-round restatement of the ground-plane flavour
(oracle/relax.cpp), which it must reproduce bit for bit."""
import numpy as np
import pytest

from relax_fixtures import (MODEL_600, add_ori_noise, axis_angle, camera_grid, grid_5x5, planar_points, qangle, qmul,
                            ring_edges, ring_edges_tracks, rx_graph_from_edges, three_cameras)


def test_full_restatement_reproduces_the_ground_plane_restatement(oracle):
    ori, pos = three_cameras()
    edges = ring_edges(ori, pos, planar_points())
    noisy = add_ori_noise(ori, [-0.1, 0.1, 0.1])
    old = oracle.relax_ground_plane(pos, ori, MODEL_600, [0, 1, 2], noisy, edges)
    g, _ = rx_graph_from_edges(oracle, pos, ori, MODEL_600, edges)
    new = g.relax([0, 1, 2], noisy, np.arange(3), oracle.options("ORIENTATION", "GROUND_PLANE"))
    assert np.array_equal(old["orientation"], new["orientation"])
    assert old["iterations_total"] == new["iterations_total"] and old["residual_blocks"] == new["residual_blocks"]
    assert np.array_equal(old["plane"], new["surface"].arrays()["vertices"])
    # a grid with NaN (uninitialised) cameras: the one-at-a-time bootstrap of relax.cpp:52-80
    ori, pos, edges, model = camera_grid(3, 4)
    start = ori.copy()
    start[[2, 7]] = np.nan
    old = oracle.relax_ground_plane(pos, start, model, np.arange(12), start, edges)
    g, _ = rx_graph_from_edges(oracle, pos, start, model, edges)
    new = g.relax(np.arange(12), start, np.arange(len(edges)), oracle.options("ORIENTATION", "GROUND_PLANE"))
    assert np.array_equal(old["orientation"], new["orientation"]) and old["solves"] == new["solves"]


def test_measurement_3_images_plane_twice(oracle):  # test_relax.cpp:416-434
    ori, pos = three_cameras()
    edges = ring_edges(ori, pos, planar_points())
    g, _ = rx_graph_from_edges(oracle, pos, ori, MODEL_600, edges)
    q = add_ori_noise(ori, [-0.1, 0.1, 0.1])
    for _ in range(2):
        q = g.relax([0, 1, 2], q, np.arange(3), oracle.options("ORIENTATION", "GROUND_PLANE"))["orientation"]
    assert max(qangle(q[i], ori[i]) for i in range(3)) < 1e-3


def test_minimal_and_grid_mesh_construction(oracle):  # expand_mesh.cpp:17-380
    ori, pos, edges, model = camera_grid(3, 4)
    s = oracle.rebuild_mesh(pos, minimal=True).arrays()
    v, e = s["vertices"], s["edges"]
    assert len(v) == 4 and len(e) == 5 and int(e[:, 2].sum()) == 4
    # border = 2 x median height; no previous surface: height = median nearest-camera distance
    d = np.sqrt(np.sort([np.sort(np.sum((pos[:, :2] - p[:2]) ** 2, axis=1))[1] for p in pos])[len(pos) // 2])
    assert np.isclose(v[0, 0], pos[:, 0].min() - 2 * d) and np.isclose(v[3, 1], pos[:, 1].max() + 2 * d)
    assert np.allclose(v[:, 2], pos[0, 2] - d)
    big = oracle.rebuild_mesh(pos, minimal=False)
    a = big.arrays()
    rows = int(np.ceil((np.ptp(pos[:, 1]) + 4 * d) / d)) + 1
    cols = int(np.ceil((np.ptp(pos[:, 0]) + 4 * d) / d)) + 1
    assert len(a["vertices"]) == rows * cols
    assert len(a["edges"]) == (rows - 1) * cols + rows * (cols - 1) + (rows - 1) * (cols - 1)
    # every interior point is found by the walk, in a triangle that contains it
    rng = np.random.default_rng(0)
    for _ in range(200):
        x = rng.uniform(a["vertices"][:, 0].min() + 1e-3, a["vertices"][:, 0].max() - 1e-3)
        y = rng.uniform(a["vertices"][:, 1].min() + 1e-3, a["vertices"][:, 1].max() - 1e-3)
        t, tri, steps = big.triangle_at(x, y)
        assert t == 2 and steps <= 100  # INTERSECTION
        P = a["vertices"][tri.astype(int)][:, :2]
        cr = lambda u, w: u[0] * w[1] - u[1] * w[0]
        sgn = [cr(P[(i + 1) % 3] - P[i], np.array([x, y]) - P[i]) for i in range(3)]
        assert all(s >= -1e-9 for s in sgn) or all(s <= 1e-9 for s in sgn)
    assert big.triangle_at(a["vertices"][:, 0].max() + 5, 0)[0] == 3  # OUTSIDE_BORDER


@pytest.mark.parametrize("minimal", [True, False])
def test_ground_mesh_three_cameras(oracle, minimal):
    """{ORIENTATION, GROUND_MESH} on the 3-camera fixture of test_relax.cpp:436-463 without the lens part: every point
    is seen by all three cameras, so the problem is built from 3-ray track blocks."""
    ori, pos = three_cameras()
    edges = ring_edges_tracks(ori, pos, planar_points())
    g, _ = rx_graph_from_edges(oracle, pos, ori, MODEL_600, edges)
    q = add_ori_noise(ori, [-0.1, 0.1, 0.1])
    opts = oracle.options("ORIENTATION", "GROUND_MESH", *(["MINIMAL_MESH"] if minimal else []))
    r = g.relax([0, 1, 2], q, np.arange(3), opts)
    assert r["track_blocks"] > 20 and r["solves"] == 2
    e0 = max(qangle(q[i], ori[i]) for i in range(3))
    e1 = max(qangle(r["orientation"][i], ori[i]) for i in range(3))
    assert e1 < 0.1 * e0, (e0, e1)
    prev = r["surface"]
    a = prev.arrays()
    assert len(a["cloud"]) > 50 and len(a["vertices"]) == (4 if minimal else len(a["vertices"]))
    # second run re-uses the previous mesh (relax_problem.cpp:1270-1274) and tightens the result
    r2 = g.relax([0, 1, 2], r["orientation"], np.arange(3), opts, prev=prev)
    assert len(r2["surface"].arrays()["vertices"]) == len(a["vertices"])
    e2 = max(qangle(r2["orientation"][i], ori[i]) for i in range(3))
    assert e2 < 5e-3, e2


def test_ground_mesh_grid_two_ray_fallback(oracle):
    """Pairwise-only features (no feature is shared between edges): no tracks, every block is a 2-ray block on the
    mesh triangle under its ray intersection, plus the mesh priors."""
    ori, pos, edges, model = camera_grid(3, 4)
    g, _ = rx_graph_from_edges(oracle, pos, ori, model, edges)
    rng = np.random.default_rng(5)
    q = np.array([qmul(ori[i], axis_angle(rng.normal(size=3) / 2, 0.05)) for i in range(len(ori))])
    r = g.relax(np.arange(12), q, np.arange(len(edges)), oracle.options("ORIENTATION", "GROUND_MESH", "MINIMAL_MESH"))
    assert r["track_blocks"] == 0 and r["two_ray_blocks"] > 100
    # 5 flat priors + 4 anchors + 1 smoothness prior on the 2-triangle mesh
    assert r["residual_blocks"] == r["two_ray_blocks"] + 10
    e0 = np.median([qangle(q[i], ori[i]) for i in range(12)])
    e1 = np.median([qangle(r["orientation"][i], ori[i]) for i in range(12)])
    assert e1 < 0.05 * e0, (e0, e1)


def _incremental(oracle, disturbed, noise, depth):
    ori, pos, edges = grid_5x5()
    start = ori.copy()
    for c in disturbed:
        start[c] = qmul(start[c], axis_angle([0, 1, 0], noise))
    g, _ = rx_graph_from_edges(oracle, pos, start, MODEL_600, edges)
    knn = oracle.knn10_bruteforce(pos[:, :2])
    r = g.relax_group(disturbed, knn, depth, oracle.options("ORIENTATION", "GROUND_PLANE"))
    return ori, start, g.orientations(), r, edges


def test_incremental_relax_center_camera(oracle):  # test_relax.cpp:833-879
    ori, start, out, r, edges = _incremental(oracle, [12], 0.2, 2)
    assert sum(1 for e in edges if 12 in (e["src"], e["dst"])) >= 8
    e0, e1 = qangle(start[12], ori[12]), qangle(out[12], ori[12])
    assert e0 > 0.1 and e1 < 0.5 * e0 and e1 < 0.05
    assert 12 in r["local_nodes"]


def test_incremental_relax_row_of_cameras(oracle):  # test_relax.cpp:925-969
    cams = [10, 11, 12, 13, 14]
    ori, start, out, r, _ = _incremental(oracle, cams, 0.15, 2)
    for c in cams:
        assert qangle(out[c], ori[c]) < qangle(start[c], ori[c]) and qangle(out[c], ori[c]) < 0.1


def test_incremental_relax_depth_3_and_large_disturbance(oracle):  # test_relax.cpp:971-1050
    ori, start, out, r, _ = _incremental(oracle, [12], 0.2, 3)
    assert qangle(out[12], ori[12]) < 0.05 and len(set(r["local_nodes"].tolist())) > 11
    ori, start, out, r, _ = _incremental(oracle, [12], 0.3, 2)
    e0, e1 = qangle(start[12], ori[12]), qangle(out[12], ori[12])
    assert e0 > 0.25 and e1 < 0.3 * e0 and e1 < 0.1


def test_relax_group_context_and_duplicates(oracle):
    """RelaxGroup::init (relax_group.cpp:14-111): depth 0 keeps only edges between primary nodes; with depth 2 the
    round-0 context nodes are appended again in round 1 (the reference rebuilds newly_connected from all of
    _directly_connected), so _local_poses holds them twice."""
    ori, pos, edges = grid_5x5()
    g, _ = rx_graph_from_edges(oracle, pos, ori, MODEL_600, edges)
    knn = oracle.knn10_bruteforce(pos[:, :2])
    r0 = g.relax_group([12, 13], knn, 0, oracle.options("ORIENTATION", "GROUND_PLANE"), run=False)
    assert sorted(r0["local_nodes"].tolist()) == [12, 13] and len(r0["opt_edges"]) == 1
    r2 = g.relax_group([12], knn, 2, oracle.options("ORIENTATION", "GROUND_PLANE"), run=False)
    nodes = r2["local_nodes"].tolist()
    first_ring = {e["dst"] if e["src"] == 12 else e["src"] for e in edges if 12 in (e["src"], e["dst"])} & set(knn[12].tolist())
    assert nodes.count(12) == 1 and all(nodes.count(c) == 2 for c in first_ring)
    assert len(r2["opt_edges"]) == len(first_ring)
