"""Pins the restated lens-model conversion (src/distort/invert_distortion.cpp:105-191 -> oracle/relax_full.cpp convertModel*)
with the reference's own tests (test/test_distort.cpp:94-137) and the intrinsics flavour of the mesh relax with
test/test_relax.cpp:436-463 (measurement_3_images_mesh_radial)."""
import numpy as np
import pytest

from relax_fixtures import (MODEL_600, add_ori_noise, planar_points, qangle, ring_edges_tracks, rx_graph_from_edges,
                            three_cameras)


def _model(radial=(0, 0, 0), tangential=(0, 0)):
    return np.array([6000.0, 2000, 1500, *radial, *tangential, 4000, 3000])


def _inverse_to_pixels(rays, inv):
    """image_from_3d(ray, InverseDifferentiableCameraModel) (distort_keypoints.cpp:105-141): the undistorted point d with
    undistort(d) = ray.xy / ray.z, by Newton iteration (the reference uses ceres::TinySolver to 1/100 px)."""
    k, p = inv[3:6], inv[6:8]
    target = rays[:, :2] / np.maximum(rays[:, 2:3], 1e-3)
    d = target.copy()

    def f(d):
        r2 = np.sum(d * d, axis=1, keepdims=True)
        rad = 1 + k[0] * r2 + k[1] * r2 ** 2 + k[2] * r2 ** 3
        prod = d[:, :1] * d[:, 1:2]
        return rad * d + 2 * prod * p[None, :] + p[None, ::-1] * (r2 + 2 * d * d)

    for _ in range(50):
        h = 1e-7
        J = np.zeros((len(d), 2, 2))
        for c in range(2):
            e = np.zeros(2)
            e[c] = h
            J[:, :, c] = (f(d + e) - f(d - e)) / (2 * h)
        d = d - np.linalg.solve(J, (f(d) - target)[:, :, None])[:, :, 0]
    return d * inv[0] + inv[1:3]


def _grid(model):
    return np.array([[i, j] for i in range(0, int(model[8]), int(model[8]) // 20) for j in range(0, int(model[9]), int(model[9]) // 20)], float)


def _verify_same(oracle, forward, inverse, eps):
    px = _grid(forward)
    # forward -> inverse: rays of the forward model land on the same pixels through the inverse model
    rays = oracle.image_to_3d(px, forward)
    assert np.max(np.abs(_inverse_to_pixels(rays, inverse) - px)) < eps
    # inverse -> forward
    rays = oracle.image_to_3d_inverse(px, inverse)
    assert np.max(np.abs(oracle.image_from_3d(rays, forward) - px)) < eps


def test_model_conversion_no_distortion(oracle):  # test_distort.cpp:94-106
    m = _model()
    inv = oracle.convert_model(m, to_inverse=True)
    assert np.array_equal(inv, m)
    _verify_same(oracle, m, inv, 1e-9)


def test_model_conversion_to_inversemodel_radial_distortion(oracle):  # test_distort.cpp:108-121
    m = _model((0.02, -0.07, 0.1))
    inv = oracle.convert_model(m, to_inverse=True)
    assert np.all(inv[3:6] != 0) and np.sign(inv[3]) == -1
    _verify_same(oracle, m, inv, 1e-2)


def test_inversemodel_conversion_to_model_radial_distortion(oracle):  # test_distort.cpp:123-137
    inv = _model((0.02, -0.07, 0.1))
    fwd = oracle.convert_model(inv, to_inverse=False)
    _verify_same(oracle, fwd, inv, 1e-2)


def test_measurement_3_images_mesh_radial(oracle):  # test_relax.cpp:436-463
    """Ten relaxes with {ORIENTATION, LENS_DISTORTIONS_RADIAL (Brown k1 k2 k3), GROUND_MESH}: the shared-intrinsics
    functors on 3-ray tracks, the inverse lens model as parameter blocks, the monotonicity cost, the model copied back
    through the forward fit after every solve.  The measurements carry no distortion (the test sets it on the group's
    copy of the model only while it generates them from the graph's model), so the radial terms must stay small."""
    ori, pos = three_cameras()
    edges = ring_edges_tracks(ori, pos, planar_points())
    g, _ = rx_graph_from_edges(oracle, pos, ori, MODEL_600, edges)
    g.persist_cam_models()
    q = add_ori_noise(ori, [-0.1, 0.1, 0.1])
    opts = oracle.options("ORIENTATION", "LENS_DISTORTIONS_RADIAL", "BROWN246", "GROUND_MESH")
    e0 = max(qangle(q[i], ori[i]) for i in range(3))
    for it in range(10):
        r = g.relax([0, 1, 2], q, np.arange(3), opts)
        q = r["orientation"]
        if it == 0:
            assert r["track_blocks"] > 20
            # tracks (no loss) + mesh priors + the monotonicity block
            assert r["residual_blocks"] > r["track_blocks"] + 5
    for i in range(3):
        assert qangle(q[i], ori[i]) < 0.1
    assert max(qangle(q[i], ori[i]) for i in range(3)) < e0
    model = r["models"][42]
    assert np.linalg.norm(model[3:6] - np.array([0.1, -0.1, 0.1])) < 0.2   # the reference's (weak) bound
    assert np.all(np.isfinite(model))
