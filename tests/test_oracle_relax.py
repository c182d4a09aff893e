"""Pins the relax restatement (oracle/relax*.cpp, mini_ceres) with restated reference tests:
test/test_relax.cpp:169-188 (downwards prior), :416-434 (measurement_3_images_plane),
:1052-1096 (robustCentroid) and finite-difference checks of the restated autodiff."""
import numpy as np
import pytest

from relax_fixtures import (DOWN, MODEL_600, add_ori_noise, axis_angle, camera_grid, planar_points, qangle, qmul,
                            ring_edges, three_cameras)


def test_downwards_prior_cost_function(oracle):  # test_relax.cpp:169-188
    q = qmul(np.array([0, 0, 0, 1.0]), axis_angle([1, 0, 0], np.pi))
    assert abs(oracle.points_downwards_prior(q, 1e-3)) < 1e-8
    q = qmul(q, axis_angle([1, 0, 0], 0.3))
    assert abs(oracle.points_downwards_prior(q, 1e-3) - 0.3e-3) < 1e-9


def test_robust_centroid(oracle):  # test_relax.cpp:1052-1096
    assert np.allclose(oracle.robust_centroid([[1, 2, 3]] * 3, 1.0), [1, 2, 3], atol=1e-6)
    r = oracle.robust_centroid([[0, 0, 0], [0.01, 0, 0], [0, 0.01, 0]], 100.0)
    assert np.linalg.norm(r - np.array([0.01 / 3, 0.01 / 3, 0])) < 0.01
    r = oracle.robust_centroid([[0, 0, 0], [1, 0, 0], [2, 0, 0], [100, 0, 0]], 1.0)
    assert abs(r[0] - 1.0) < 0.5 and abs(r[1]) < 1e-6 and abs(r[2]) < 1e-6
    r = oracle.robust_centroid([[0, 0, 0], [2, 0, 0]], 10.0)
    assert abs(r[0] - 1.0) < 1e-4 and abs(r[1]) < 1e-6
    assert np.allclose(oracle.robust_centroid([[5, 3, 1]], 1.0), [5, 3, 1], atol=1e-6)


def test_plane_intersection_cost_zero_at_truth_and_autodiff_matches_finite_differences(oracle):
    ori, pos = three_cameras()
    p = np.array([8.0, 9.5, -10 + 1e-3 * 3 + 1e-2 * 4.5])
    plane_xy = np.array([[-40.0, -40], [60, -40], [10, 60]])
    z = np.array([-10 + 1e-3 * (x - 5) + 1e-2 * (y - 5) for x, y in plane_xy])
    from relax_fixtures import qinv, qrot

    rays = np.array([qrot(qinv(ori[i]), (p - pos[i]) / np.linalg.norm(p - pos[i])) for i in range(2)])
    ok, res, jac = oracle.plane_intersection_cost(pos[:2], rays, plane_xy, ori[0], ori[1], z)
    assert ok and np.max(np.abs(res)) < 1e-12
    # perturbed state: autodiff Jacobian vs central differences
    q0 = qmul(ori[0], axis_angle([0, 1, 0], 0.05))
    q1 = qmul(ori[1], axis_angle([1, 0, 0], -0.03))
    z2 = z + np.array([0.3, -0.2, 0.1])
    ok, res, jac = oracle.plane_intersection_cost(pos[:2], rays, plane_xy, q0, q1, z2)
    assert ok and np.max(np.abs(res)) > 1e-4
    x0 = np.concatenate([q0, q1, z2])
    for c in range(11):
        h = 1e-6
        xp, xm = x0.copy(), x0.copy()
        xp[c] += h
        xm[c] -= h
        rp = oracle.plane_intersection_cost(pos[:2], rays, plane_xy, xp[:4], xp[4:8], xp[8:], False)[1]
        rm = oracle.plane_intersection_cost(pos[:2], rays, plane_xy, xm[:4], xm[4:8], xm[8:], False)[1]
        assert np.allclose((rp - rm) / (2 * h), jac[:, c], rtol=1e-5, atol=1e-7), c


def test_measurement_3_images_plane(oracle):  # test_relax.cpp:416-434
    ori, pos = three_cameras()
    edges = ring_edges(ori, pos, planar_points())
    noisy = add_ori_noise(ori, [-0.1, 0.1, 0.1])
    r = oracle.relax_ground_plane(pos, ori, MODEL_600, [0, 1, 2], noisy, edges)
    r = oracle.relax_ground_plane(pos, ori, MODEL_600, [0, 1, 2], r["orientation"], edges)  # "and again"
    for i in range(3):
        assert qangle(r["orientation"][i], ori[i]) < 1e-3
    assert r["final_cost"] < r["initial_cost"] or r["initial_cost"] < 1e-12
    assert r["residual_blocks"] > 3


def test_exact_start_converges_immediately(oracle):
    """Ceres behaviour the reference asserts elsewhere (test_relax.cpp:513-519): starting at the optimum
    the solver needs at most a couple of iterations."""
    ori, pos = three_cameras()
    edges = ring_edges(ori, pos, planar_points())
    r = oracle.relax_ground_plane(pos, ori, MODEL_600, [0, 1, 2], ori, edges)
    for i in range(3):
        assert qangle(r["orientation"][i], ori[i]) < 1e-6


def test_nan_orientations_are_bootstrapped_one_at_a_time(oracle):
    """runGroundPlane (relax.cpp:51-80): NaN poses start from the previous node's orientation and get
    their own solve each."""
    ori, pos, edges, model = camera_grid(2, 3, seed=4)
    start = ori.copy()
    start[:] = np.nan
    r = oracle.relax_ground_plane(pos, ori, model, np.arange(6), start, edges)
    assert r["solves"] >= 2 * 6
    for i in range(6):
        assert qangle(r["orientation"][i], ori[i]) < 0.2   # yaw is only weakly observable from a plane


def test_grid_recovers_orientations(oracle):
    ori, pos, edges, model = camera_grid(3, 4, seed=7)
    rng = np.random.default_rng(0)
    noisy = np.array([qmul(q, axis_angle(rng.normal(size=3) / 1.7, 0.1)) for q in ori])
    before = max(qangle(noisy[i], ori[i]) for i in range(len(ori)))
    r = oracle.relax_ground_plane(pos, ori, model, np.arange(len(ori)), noisy, edges)
    r = oracle.relax_ground_plane(pos, ori, model, np.arange(len(ori)), r["orientation"], edges)
    after = max(qangle(r["orientation"][i], ori[i]) for i in range(len(ori)))
    assert after < before * 0.2
