"""The two relax flavours only the reference's tests reach, restated in the oracle (oracle/relax_full.cpp:
setupDecompositionProblem + MultiDecomposedRotationCost, setup3dPointProblem + PixelErrorCost_*, the refined two-view
triangulation of intersection.cpp:163-186) and held against the reference's own expectations: test/test_relax.cpp:298-414
(relative orientation) and :485-682 (3-D points) - including the ITERATION COUNTS and cost levels those tests assert of
Ceres, which is the tightest pin of the restated trust-region LM the reference offers."""
import numpy as np

from oracle import pyoracle as oracle
from relax_fixtures import DOWN, MODEL_600, add_ori_noise, axis_angle, project, qangle, qinv, qmul, qrot, ring_edges_tracks, three_cameras

ORI = oracle.options("ORIENTATION")
PTS = oracle.options("ORIENTATION", "POINTS_3D")


def points_3d():
    """generate_3d_points(): test_relax.cpp:75-89"""
    return np.array([[i + 5, j + 5, -10 + (i + j) % 2] for i in range(10) for j in range(10)], float)


def _graph(ori, pos, model=MODEL_600, graph_ori=None):
    g = oracle.RxGraph()
    g.add_model(model, 42)
    for i in range(len(ori)):
        g.add_node(pos[i], (ori if graph_ori is None else graph_ori)[i])
    return g


def _point_edges(g, ori, pos, points, model=MODEL_600):
    """add_point_measurements(): features appended per camera, edge i -> (i + 1) % 3 over all points (:91-125)"""
    edges = ring_edges_tracks(ori, pos, points, model)
    ids = []
    for e in edges:
        k = np.arange(len(points))
        ids.append(g.add_edge(e["src"], e["dst"], e["px"], k, k, match_index=k))
    return ids


def _nearest_sq(points, truth):
    return np.array([np.min(np.sum((truth - p) ** 2, axis=1)) for p in points])


def test_no_images_and_single_prior():
    g = oracle.RxGraph()
    out = g.relax(np.zeros(0, np.uint64), np.zeros((0, 4)), [], ORI)                  # no_images (:298-309): no crash
    assert out["solves"] == 0
    # prior_1_image (:311-335): tilted 45 degrees, the downward prior pulls it back
    q0 = axis_angle([1, 0, 0], np.pi / 4)
    g = _graph([q0], [np.array([9.0, 9, 9])])
    out = g.relax([0], [q0], [], ORI)
    assert qangle(out["orientation"][0], DOWN) < np.pi / 4 and out["solves"] == 1


def test_prior_2_images():
    """:337-377: relative pose identity along x, cameras 135 degrees apart about y -> relative orientation -> identity"""
    ori = [axis_angle([0, 1, 0], np.pi / 2), axis_angle([0, 1, 0], -np.pi / 4)]
    pos = [np.array([9.0, 9, 9]), np.array([11.0, 9, 9])]
    g = _graph(ori, pos)
    poses = np.full((4, 8), np.nan)
    poses[:, 7] = 0
    poses[0] = [0, 0, 0, 1, 1, 0, 0, 8]
    e = g.add_edge(0, 1, np.zeros((10, 4)), np.arange(10), np.arange(10), poses=poses)
    out = g.relax([0, 1], np.array(ori), [e], ORI)
    rel = qmul(qinv(out["orientation"][0]), out["orientation"][1])
    assert qangle(rel, np.array([0, 0, 0, 1.0])) < 1e-3


def test_relative_orientation_3_images():
    """:379-396: the decompositions of add_edge_measurements (:127-149), orientations disturbed by a radian each"""
    ori, pos = three_cameras()
    g = _graph(ori, pos)
    ids = []
    for i in range(3):
        a, b = i, (i + 1) % 3
        r = qmul(ori[b], qinv(ori[a]))
        d = pos[b] - pos[a]
        t = qrot(qinv(ori[a]), d / np.linalg.norm(d))
        poses = np.full((4, 8), np.nan)
        poses[:, 7] = 0
        if i in (0, 2):
            poses[0] = [*r, *t, 8]
        if i in (1, 2):
            poses[1] = [*r, *t, 18]
        ids.append(g.add_edge(a, b, np.zeros((1, 4)), [0], [0], poses=poses))
    start = add_ori_noise(ori, [-1, 1, 1])
    out = g.relax([0, 1, 2], start, ids, ORI)
    assert max(qangle(out["orientation"][i], ori[i]) for i in range(3)) < 1e-5


def test_measurement_3_images_points():
    """:398-413: reprojection bundle of 100 points seen by all three cameras puts them back to 1e-8"""
    ori, pos = three_cameras()
    g = _graph(ori, pos)
    ids = _point_edges(g, ori, pos, points_3d())
    start = add_ori_noise(ori, [-0.05, 0.05, 0.05])
    out = g.relax([0, 1, 2], start, ids, PTS)
    assert max(qangle(out["orientation"][i], ori[i]) for i in range(3)) < 1e-8
    assert out["solves"] == 2                       # relaxObservedModelOnly, then the joint solve (relax.cpp:104-115)


def test_point_triangulation_exact():
    """:485-519: exact measurements -> the set-up triangulates to 1e-8 (squared), the solver stops after <= 2 iterations
    with initial and final cost below 1e-10"""
    ori, pos = three_cameras()
    truth = points_3d()
    noisy_graph = add_ori_noise(ori, [-0.1, 0.1, 0.1])          # add_ori_noise_graph: the graph's orientations are unused
    g = _graph(ori, pos, graph_ori=noisy_graph)
    ids = _point_edges(g, ori, pos, truth)
    out = g.points_problem([0, 1, 2], ori, ids, PTS, mode=1)
    assert 200 < len(out["points_before"]) <= 300           # the 0.05 grid filter keeps the best inlier per cell
    assert _nearest_sq(out["points_before"], truth).max() < 1e-8
    assert out["iterations"] <= 2 and out["initial_cost"] < 1e-10 and out["final_cost"] < 1e-10


def test_point_triangulation_noise():
    """:521-557: with 0.05 rad of orientation noise the triangulated points start > 1 (squared) off, the solve brings them
    to 1e-8; it takes MORE THAN 10 iterations, starts above 4e2 and ends below 1e-10 - Ceres' numbers for this problem"""
    ori, pos = three_cameras()
    truth = points_3d()
    g = _graph(ori, pos)
    ids = _point_edges(g, ori, pos, truth)
    start = add_ori_noise(ori, [-0.05, 0.05, 0.05])
    out = g.points_problem([0, 1, 2], start, ids, PTS, mode=1)
    assert _nearest_sq(out["points_before"], truth).min() > 1
    assert _nearest_sq(out["points_after"], truth).max() < 1e-8
    assert out["iterations"] > 10 and out["initial_cost"] > 4e2 and out["final_cost"] < 1e-10
    assert max(qangle(out["orientation"][i], ori[i]) for i in range(3)) < 1e-8


def test_point_triangulation_focal_principal():
    """:597-637: measurements made with f x 0.8 and pp (380, 320); the solve from (f, (400, 300)) runs and moves the model
    towards them (the reference's own tolerances: 100 px on f, 50 px on the principal point)"""
    ori, pos = three_cameras()
    truth = points_3d()
    measured = MODEL_600.copy()
    measured[0] *= 0.8
    measured[1:3] = [380, 320]
    g = _graph(ori, pos)
    ids = _point_edges(g, ori, pos, truth, measured)
    start = add_ori_noise(ori, [-0.05, 0.05, 0.05])
    out = g.points_problem([0, 1, 2], start, ids, oracle.options("ORIENTATION", "POINTS_3D", "FOCAL_LENGTH", "PRINCIPAL_POINT"), mode=1,
                           model10=MODEL_600)
    assert out["iterations"] > 0
    assert abs(out["model"][0] - measured[0]) < 100 and abs(out["model"][1] - 380) < 50 and abs(out["model"][2] - 320) < 50


def test_point_triangulation_accuracy():
    """:639-682: relaxObservedModelOnly moves fewer than 30 of the ~290 badly triangulated points by more than 0.1"""
    ori, pos = three_cameras()
    truth = points_3d()
    g = _graph(ori, pos)
    ids = _point_edges(g, ori, pos, truth)
    start = add_ori_noise(ori, [-0.05, 0.05, 0.05])
    out = g.points_problem([0, 1, 2], start, ids, PTS, mode=2)
    assert _nearest_sq(out["points_before"], truth).min() > 1
    moved = np.linalg.norm(out["points_before"] - out["points_after"], axis=1) > 0.1
    assert moved.sum() < 30
