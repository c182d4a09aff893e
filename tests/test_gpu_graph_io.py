"""A graph linked on the device, written as graph.json, loaded again and relaxed: the loaded graph is a full input of the
relax stage (what SURVEY.md §8 f3 wants the format for), and a checkpoint of it resumes with the same result."""
import numpy as np
import pytest

from opencalibration_amd import capi, host, synth
from oracle import graph_json as gj
from relax_fixtures import qangle
from test_gpu_pipeline import perturbed

pytestmark = pytest.mark.gpu


def test_linked_graph_survives_the_file(tmp_path):
    grid = synth.make_grid(**synth.CONFIGS["C1"])
    ctx = capi.Context(0)
    g = host.Graph.from_synthetic(grid)
    g.link(ctx)
    start = perturbed(grid.orientation, 0.1, 1)
    g.set_orientations(start)
    text = g.to_json()
    doc = gj.read_graph(text)                                    # the independent reader sees the device's edges
    edges = g.edges(with_distances=True)
    assert len(doc["edges"]) == len(edges) == 90
    for e in edges:
        d = [x for x in doc["edges"].values() if (x["source"], x["dest"]) == (e["source"], e["dest"])][0]
        assert np.array_equal(d["relation"], e["H"].reshape(9)) and len(d["inlier_matches"]) == e["n_inliers"]
        assert np.array_equal([m[2] for m in d["matches"]], e["dist"])
    host.save_checkpoint(tmp_path / "cp", g, state="INITIAL_PROCESSING", state_run_count=1)
    g2, surfaces, meta = host.load_checkpoint(tmp_path / "cp")
    assert g2.to_json() == text and meta["state_run_count"] == 1
    # relax both: the loaded graph iterates its nodes in id order, so the problem is assembled in another order
    order = [g.node_ids.index(i) for i in g2.node_ids]
    a = g.relax_ground_plane(ctx, start)
    b = g2.relax_ground_plane(ctx, start[order])
    assert int(a["residual_blocks"]) == int(b["residual_blocks"])
    worst = max(qangle(a["orientation"][order[i]], b["orientation"][i]) for i in range(grid.n_images))
    assert worst < 1e-6, worst
    err = max(qangle(b["orientation"][i], grid.orientation[order[i]]) for i in range(grid.n_images))
    assert err < 5e-3, err
    # and the relaxed orientations are what the next file holds
    g2.set_orientations(b["orientation"])
    back = gj.read_graph(g2.to_json())
    assert np.array_equal([back["nodes"][i]["orientation"] for i in g2.node_ids], b["orientation"])
    g.close(), g2.close(), ctx.close()
