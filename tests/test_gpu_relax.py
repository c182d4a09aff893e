"""GPU parity of the relax step (ground-plane flavour) against the oracle, through liboc_host.so
(relax_ground_plane -> ochip_relax_*).  Floating point: tolerance 1e-6 rad on poses as BASELINE.json
states ("relax-stage pose deltas within 1e-6 of reference"); in practice the two LM trajectories agree
to ~1e-10 and use the same number of iterations."""
import numpy as np
import pytest

from opencalibration_amd import capi, host
from relax_fixtures import (MODEL_600, add_ori_noise, axis_angle, camera_grid, planar_points, qangle, qmul, ring_edges,
                            three_cameras)

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-6


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _both(ctx, oracle, pos, ori, model, pose_node, start, edges):
    exp = oracle.relax_ground_plane(pos, ori, model, pose_node, start, edges)
    got = host.relax_ground_plane(ctx, pos, ori, model, pose_node, start, oracle.pack_edges(edges))
    return exp, got


def _assert_same(exp, got):
    n = len(exp["orientation"])
    worst = max(qangle(exp["orientation"][i], got["orientation"][i]) for i in range(n))
    assert worst < POSE_TOL, worst
    assert np.allclose(exp["plane"], got["plane"], rtol=0, atol=1e-5)
    assert got["residual_blocks"] == exp["residual_blocks"]
    assert got["solves"] == exp["solves"]
    assert abs(got["iterations_total"] - exp["iterations_total"]) <= 2
    assert got["final_cost"] == pytest.approx(exp["final_cost"], rel=1e-6, abs=1e-14)


def test_measurement_3_images_plane(ctx, oracle):  # test/test_relax.cpp:416-434 on the device
    ori, pos = three_cameras()
    edges = ring_edges(ori, pos, planar_points())
    noisy = add_ori_noise(ori, [-0.1, 0.1, 0.1])
    exp, got = _both(ctx, oracle, pos, ori, MODEL_600, [0, 1, 2], noisy, edges)
    _assert_same(exp, got)
    exp2, got2 = _both(ctx, oracle, pos, ori, MODEL_600, [0, 1, 2], got["orientation"], edges)
    _assert_same(exp2, got2)
    for i in range(3):
        assert qangle(got2["orientation"][i], ori[i]) < 1e-3


def test_context_cameras_stay_constant(ctx, oracle):
    """Only camera 1 is optimised; 0 and 2 come from the graph as constant context (nodeid2poseopt)."""
    ori, pos = three_cameras()
    edges = ring_edges(ori, pos, planar_points())
    start = qmul(ori[1], axis_angle([0, 0, 1], 0.1))[None, :]
    exp, got = _both(ctx, oracle, pos, ori, MODEL_600, [1], start, edges)
    _assert_same(exp, got)
    assert qangle(got["orientation"][0], ori[1]) < 1e-3


def test_nan_bootstrap(ctx, oracle):
    ori, pos, edges, model = camera_grid(2, 3, seed=4)
    start = np.full_like(ori, np.nan)
    exp, got = _both(ctx, oracle, pos, ori, model, np.arange(6), start, edges)
    _assert_same(exp, got)


@pytest.mark.parametrize("rows,cols", [(3, 4), (6, 8)])
def test_grid_parity(ctx, oracle, rows, cols):
    ori, pos, edges, model = camera_grid(rows, cols, seed=7)
    rng = np.random.default_rng(0)
    noisy = np.array([qmul(q, axis_angle(rng.normal(size=3) / 1.7, 0.1)) for q in ori])
    exp, got = _both(ctx, oracle, pos, ori, model, np.arange(len(ori)), noisy, edges)
    _assert_same(exp, got)
    assert max(qangle(got["orientation"][i], ori[i]) for i in range(len(ori))) < 1e-6


def test_large_system_uses_blocked_cholesky(ctx, oracle):
    """> 64 tangent dimensions per panel: 10 x 12 cameras = 363 unknowns exercises the multi-panel
    Cholesky and the blocked triangular solves; checked against the oracle's dense solve."""
    ori, pos, edges, model = camera_grid(10, 12, seed=11, pts_per_side=30)
    rng = np.random.default_rng(1)
    noisy = np.array([qmul(q, axis_angle(rng.normal(size=3) / 1.7, 0.05)) for q in ori])
    exp, got = _both(ctx, oracle, pos, ori, model, np.arange(len(ori)), noisy, edges)
    _assert_same(exp, got)


def test_shuffled_camera_order(ctx, oracle):
    """The cameras arrive in random order: the unknowns are renumbered (reverse Cuthill-McKee) before the envelope
    factorisation, and the result is still the oracle's dense solve of the same shuffled problem."""
    ori, pos, edges, model = camera_grid(10, 12, seed=5, pts_per_side=30)
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(ori))
    inv = np.argsort(perm)
    ori, pos = ori[perm], pos[perm]
    edges = [dict(e, src=int(inv[e["src"]]), dst=int(inv[e["dst"]])) for e in edges]
    noisy = np.array([qmul(q, axis_angle(rng.normal(size=3) / 1.7, 0.05)) for q in ori])
    exp, got = _both(ctx, oracle, pos, ori, model, np.arange(len(ori)), noisy, edges)
    _assert_same(exp, got)


def test_graph_without_edges(ctx, oracle):
    """No measurements: only the PointsDownwardsPrior blocks remain (relax_problem.cpp:1297), the cameras barely
    move and the device agrees with the oracle on that degenerate problem too."""
    from opencalibration_amd import host, synth

    grid = synth.make_grid(1, 3, feats=64, seed=2)
    g = host.Graph.from_synthetic(grid)     # never linked: no edges
    got = g.relax_ground_plane(ctx, grid.orientation)
    exp = oracle.relax_ground_plane(grid.position, grid.orientation, grid.model, np.arange(3), grid.orientation, [])
    assert int(got["residual_blocks"]) == exp["residual_blocks"]
    dots = np.abs(np.sum(got["orientation"] * exp["orientation"], axis=1))
    assert np.all(2 * np.arccos(np.clip(dots, 0, 1)) < 1e-6)
    dots = np.abs(np.sum(got["orientation"] * grid.orientation, axis=1))
    assert np.all(2 * np.arccos(np.clip(dots, 0, 1)) < 1e-2)
    g.close()


def test_tile_factorisation_equals_the_launch_chain():
    """OCHIP_TEST_HOOKS=chol_verify factors every reduced system twice - the one-launch tile Cholesky (workgroups handing tiles to
    each other) and the chain of dependent launches - and fails the solve when the forward solves differ by more than
    1e-7 relative (1e-9 until the diagonal tile was blocked: nearly singular 12-unknown systems differ by 2e-9).  Run in a child process (the switch is read once) over a plane problem with the augmented row inside
    the last diagonal tile (n % 64 != 0), one with n % 64 == 0, and a mesh problem with a dense tail."""
    import os
    import subprocess
    import sys

    code = r'''
import sys
sys.path.insert(0, "tests")
import numpy as np
from opencalibration_amd import capi, host
from relax_fixtures import camera_grid, camera_grid_tracks, pack_edges_with_features
from oracle import pyoracle
ctx = capi.Context(0)
for rows, cols in ((5, 6), (3, 7), (8, 8)):          # 90 + 3 = 93, 63 + 3 = 66, 192 + 3 = 195 unknowns
    ori, pos, edges, model = camera_grid(rows, cols)
    rng = np.random.default_rng(rows)
    start = ori + rng.normal(0, 0.02, ori.shape)
    start /= np.linalg.norm(start, axis=1, keepdims=True)
    out = host.relax_ground_plane(ctx, pos, ori, model, np.arange(len(pos)), start, host.pack_edges(edges))
    assert out["iterations_total"] > 3, out
ori, pos, edges, model = camera_grid_tracks(6, 7)
pk, feats = pack_edges_with_features(pyoracle, len(pos), edges)
gx = np.linspace(-4, 16, 9)
prev = host.Surface().set(np.zeros((0, 3)), np.zeros((0, 5), np.uint64), np.array([[x, y, 1e-3 * x + 1e-2 * y] for x in gx for y in gx]))
out = host.relax(ctx, pos, ori, model, feats, np.arange(len(pos)), ori, pk, host.relax_options("ORIENTATION", "GROUND_MESH"), 0.1, previous=prev)
assert out["unknowns"] > 500 and out["iterations_total"] > 1, out
print("verified", out["unknowns"])
'''
    env = dict(os.environ, OCHIP_TEST_HOOKS="chol_verify")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "verified" in r.stdout, r.stdout + r.stderr
