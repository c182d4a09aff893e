"""GPU parity of the {ORIENTATION, GROUND_MESH} relax flavour (the general engine: csrc/relax_general.hip behind
host/relax_mesh.cpp) against the oracle's restatement (oracle/relax_full.cpp).  Poses within 1e-6 rad (BASELINE.json),
mesh heights within 1e-5, the same residual blocks (tracks, 2-ray blocks, priors) and the same number of solves."""
import numpy as np
import pytest

from opencalibration_amd import capi, host
from relax_fixtures import (MODEL_600, add_ori_noise, axis_angle, camera_grid, camera_grid_tracks, pack_edges_with_features,
                            planar_points, qangle, qmul, ring_edges_tracks, rx_graph_from_edges, three_cameras)

pytestmark = pytest.mark.gpu
POSE_TOL, Z_TOL = 1e-6, 1e-5


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _both(ctx, oracle, pos, ori, model, start, edges, opts, frac=0.1, prev=None):
    n = len(pos)
    g, _ = rx_graph_from_edges(oracle, pos, ori, model, edges)
    oprev = hprev = None
    if prev is not None:
        oprev = oracle.RxSurface().set(prev["vertices"], prev["edges"], prev["cloud"])
        hprev = host.Surface().set(prev["vertices"], prev["edges"], prev["cloud"])
    exp = g.relax(np.arange(n), start, np.arange(len(edges)), oracle.options(*opts), frac, oprev)
    pk, feats = pack_edges_with_features(oracle, n, edges)
    got = host.relax(ctx, pos, ori, model, feats, np.arange(n), start, pk, host.relax_options(*opts), frac, previous=hprev)
    return exp, got


def _assert_same(exp, got, iter_slack=2):
    n = len(exp["orientation"])
    assert got["track_blocks"] == exp["track_blocks"] and got["two_ray_blocks"] == exp["two_ray_blocks"]
    assert got["residual_blocks"] == exp["residual_blocks"] and got["solves"] == exp["solves"]
    worst = max(qangle(exp["orientation"][i], got["orientation"][i]) for i in range(n))
    assert worst < POSE_TOL, worst
    ea, ga = exp["surface"].arrays(), got["surface"].arrays()
    assert np.array_equal(ea["vertices"][:, :2], ga["vertices"][:, :2]) and np.array_equal(ea["edges"], ga["edges"])
    assert np.max(np.abs(ea["vertices"][:, 2] - ga["vertices"][:, 2])) < Z_TOL
    assert abs(got["iterations_total"] - exp["iterations_total"]) <= iter_slack
    assert got["final_cost"] == pytest.approx(exp["final_cost"], rel=1e-5, abs=1e-14)
    assert ea["cloud"].shape == ga["cloud"].shape and np.allclose(ea["cloud"], ga["cloud"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("minimal", [True, False])
def test_three_cameras_track_blocks(ctx, oracle, minimal):
    ori, pos = three_cameras()
    edges = ring_edges_tracks(ori, pos, planar_points())
    q = add_ori_noise(ori, [-0.1, 0.1, 0.1])
    opts = ["ORIENTATION", "GROUND_MESH"] + (["MINIMAL_MESH"] if minimal else [])
    exp, got = _both(ctx, oracle, pos, ori, MODEL_600, q, edges, opts)
    assert exp["track_blocks"] > 20
    _assert_same(exp, got)
    # the next run re-uses the relaxed mesh
    exp2, got2 = _both(ctx, oracle, pos, ori, MODEL_600, got["orientation"], edges, opts, prev=got["surface"].arrays())
    _assert_same(exp2, got2)
    assert max(qangle(got2["orientation"][i], ori[i]) for i in range(3)) < 5e-3


@pytest.mark.parametrize("minimal", [True, False])
def test_grid_two_ray_blocks_only(ctx, oracle, minimal):
    ori, pos, edges, model = camera_grid(3, 4)
    rng = np.random.default_rng(5)
    q = np.array([qmul(ori[i], axis_angle(rng.normal(size=3) / 2, 0.05)) for i in range(len(ori))])
    opts = ["ORIENTATION", "GROUND_MESH"] + (["MINIMAL_MESH"] if minimal else [])
    exp, got = _both(ctx, oracle, pos, ori, model, q, edges, opts)
    assert exp["track_blocks"] == 0 and exp["two_ray_blocks"] > 100
    _assert_same(exp, got)


@pytest.mark.parametrize("frac", [0.1, 0.05])
def test_grid_tracks_and_two_ray_fallback(ctx, oracle, frac):
    """Shared features with pixel noise: tracks of 3..5 rays take the cells they cover, 2-ray blocks fill in
    (relax_problem.cpp:93-115).  The grid mesh (783 vertices) takes its heights from a previous surface's point cloud;
    the second run re-uses the relaxed mesh."""
    ori, pos, edges, model = camera_grid_tracks(6, 7)
    rng = np.random.default_rng(8)
    for e in edges:
        e["px"] = e["px"] + rng.normal(0, 0.3, e["px"].shape)
    q = np.array([qmul(ori[i], axis_angle(rng.normal(size=3) / 2, 0.05)) for i in range(len(ori))])
    gx = np.linspace(-4, 16, 9)
    cloud = np.array([[x, y, 1e-3 * x + 1e-2 * y] for x in gx for y in gx])
    prev = dict(vertices=np.zeros((0, 3)), edges=np.zeros((0, 5), np.uint64), cloud=cloud)
    exp, got = _both(ctx, oracle, pos, ori, model, q, edges, ["ORIENTATION", "GROUND_MESH"], frac, prev)
    assert exp["track_blocks"] > 50 and exp["two_ray_blocks"] > 1000 and len(exp["surface"].arrays()["vertices"]) > 500
    _assert_same(exp, got, iter_slack=3)
    exp2, got2 = _both(ctx, oracle, pos, ori, model, got["orientation"], edges, ["ORIENTATION", "GROUND_MESH"], frac,
                       got["surface"].arrays())
    _assert_same(exp2, got2, iter_slack=3)
    assert np.median([qangle(got2["orientation"][i], ori[i]) for i in range(len(ori))]) < 2e-3


def test_ground_plane_through_the_general_entry_point(ctx, oracle):
    ori, pos, edges, model = camera_grid(3, 4)
    start = ori.copy()
    start[[2, 7]] = np.nan
    exp, got = _both(ctx, oracle, pos, start, model, start, edges, ["ORIENTATION", "GROUND_PLANE"])
    n = len(ori)
    assert max(qangle(exp["orientation"][i], got["orientation"][i]) for i in range(n)) < POSE_TOL
    assert got["solves"] == exp["solves"]
    assert np.allclose(exp["surface"].arrays()["vertices"], got["surface"].arrays()["vertices"], rtol=0, atol=1e-5)
