"""RelaxStage on the device (csrc/host/relax_stage.cpp: partition -> RelaxGroup per cluster -> one runner per group on sibling
device contexts -> write-back -> merged surface) against the oracle's RelaxGroup restatement driven group by group:
the incremental single-group form of INITIAL_PROCESSING (new cameras + two rings of context cameras, ground plane;
test/test_relax.cpp:833-1050) and the clustered form of the later states (floor(n / 50) groups, ground mesh)."""
import numpy as np
import pytest

from opencalibration_amd import capi, host
from relax_fixtures import (MODEL_600, axis_angle, camera_grid_tracks, grid_5x5, host_graph_from_edges, host_paths, qangle, qmul,
                            rx_graph_from_edges)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("cams,noise", [([12], 0.2), ([10, 11, 12, 13, 14], 0.15), ([12], 0.3)])
def test_incremental_relax_matches_oracle(ctx, oracle, cams, noise):
    ori, pos, edges = grid_5x5()
    start = ori.copy()
    for c in cams:
        start[c] = qmul(start[c], axis_angle([0, 1, 0], noise))
    rx, _ = rx_graph_from_edges(oracle, pos, start, MODEL_600, edges, paths=host_paths(25))
    exp = rx.relax_group(cams, oracle.knn10_bruteforce(pos[:, :2]), 2, oracle.options("ORIENTATION", "GROUND_PLANE"))
    exp_ori = rx.orientations()
    g = host_graph_from_edges(host, pos, start, MODEL_600, edges)
    ids = [g.node_ids[c] for c in cams]
    got = g.relax_stage(ctx, host.relax_options("ORIENTATION", "GROUND_PLANE"), node_ids=ids, disable_parallelism=True)
    assert int(got["groups"]) == 1 and int(got["solves"]) == exp["solves"]
    assert int(got["residual_blocks"]) == exp["residual_blocks"]
    got_ori = np.array([e for e in g.orientations()])
    worst = max(qangle(exp_ori[i], got_ori[i]) for i in range(25))
    assert worst < 1e-6, worst
    for c in cams:
        assert qangle(got_ori[c], ori[c]) < 0.1 and qangle(got_ori[c], ori[c]) < qangle(start[c], ori[c])
    g.close()


def test_clustered_mesh_relax_matches_oracle(ctx, oracle):
    ori, pos, edges, model = camera_grid_tracks(10, 12, pts_per_side=30)
    n = len(pos)
    rng = np.random.default_rng(3)
    for e in edges:
        e["px"] = e["px"] + rng.normal(0, 0.3, e["px"].shape)
    start = np.array([qmul(ori[i], axis_angle(rng.normal(size=3) / 2, 0.03)) for i in range(n)])
    gx = np.linspace(pos[:, 0].min() - 6, pos[:, 0].max() + 6, 12)
    gy = np.linspace(pos[:, 1].min() - 6, pos[:, 1].max() + 6, 12)
    cloud = np.array([[x, y, 1e-3 * x + 1e-2 * y] for x in gx for y in gy])
    prev_h = host.rebuild_mesh(pos, host.Surface().set(np.zeros((0, 3)), np.zeros((0, 5), np.uint64), cloud), minimal=True)
    pa = prev_h.arrays()
    prev_o = oracle.RxSurface().set(pa["vertices"], pa["edges"])
    g = host_graph_from_edges(host, pos, start, model, edges)
    got = g.relax_stage(ctx, host.relax_options("ORIENTATION", "GROUND_MESH"), 0.1, previous=prev_h)
    groups = got["group_of_node"]
    assert int(got["groups"]) == n // 50 == 2 and set(groups.tolist()) == {0, 1}
    got_ori = np.array([e for e in g.orientations()])
    # the oracle, group by group on the same partition
    rx, _ = rx_graph_from_edges(oracle, pos, start, model, edges, paths=host_paths(n))
    k, assign, depth = oracle.relax_stage_groups(rx)
    assert k == 2 and depth == 0 and np.array_equal(assign, groups)
    ordered, _ = oracle.relax_stage_groups(rx, ordered=True)
    assert ordered == g.relax_partition(2, ordered=True)   # the same node order inside the groups: it decides the edge order
    knn = oracle.knn10_bruteforce(pos[:, :2])
    surfaces, blocks = [], 0
    for grp in range(k):
        r = rx.relax_group(ordered[grp], knn, 0, oracle.options("ORIENTATION", "GROUND_MESH"), 0.1, prev_o)
        surfaces.append(r["surface"].arrays())
        blocks += r["residual_blocks"]
    exp_ori = rx.orientations()
    assert int(got["residual_blocks"]) == blocks
    worst = max(qangle(exp_ori[i], got_ori[i]) for i in range(n))
    assert worst < 1e-6, worst
    assert np.median([qangle(got_ori[i], ori[i]) for i in range(n)]) < 3e-3
    # merged surface: every vertex lies between the two groups' heights (a point-count weighted mean of them)
    mv = got["surface"].arrays()["vertices"]
    lo = np.minimum(surfaces[0]["vertices"][:, 2], surfaces[1]["vertices"][:, 2]) - 1e-5
    hi = np.maximum(surfaces[0]["vertices"][:, 2], surfaces[1]["vertices"][:, 2]) + 1e-5
    assert np.all(mv[:, 2] >= lo) and np.all(mv[:, 2] <= hi)
    assert len(got["surface"].arrays()["cloud"]) == len(surfaces[0]["cloud"]) + len(surfaces[1]["cloud"])
    g.close()
