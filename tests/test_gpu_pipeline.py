"""End-to-end on the device: synthetic features -> LinkStage -> relax(ground plane) on the linked graph,
checked against the oracle relax run on the same edges (which tests/test_gpu_link.py shows are bit-exact
with the oracle's own link restatement).  BASELINE config C1 (10 images x 2k features)."""
import numpy as np
import pytest

from opencalibration_amd import capi, host, synth
from relax_fixtures import axis_angle, qangle, qmul

pytestmark = pytest.mark.gpu


def perturbed(orientation, sigma, seed):
    rng = np.random.default_rng(seed)
    out = []
    for q in orientation:
        a = rng.normal(size=3)
        out.append(qmul(q, axis_angle(a / np.linalg.norm(a), sigma)))
    return np.array(out)


def test_c1_link_then_relax_matches_oracle(oracle):
    grid = synth.make_grid(**synth.CONFIGS["C1"])
    ctx = capi.Context(0)
    g = host.Graph.from_synthetic(grid)
    g.link(ctx)
    start = perturbed(grid.orientation, 0.1, 1)      # 0.1 rad initial error (test/test_relax.cpp:421)
    g.set_orientations(start)
    got = g.relax_ground_plane(ctx, start)
    edges = g.edges_flat()
    assert len(edges) == 90
    exp = oracle.relax_ground_plane(grid.position, start, grid.model, np.arange(grid.n_images), start, edges)
    worst = max(qangle(exp["orientation"][i], got["orientation"][i]) for i in range(grid.n_images))
    assert worst < 1e-6, worst
    assert int(got["residual_blocks"]) == exp["residual_blocks"]
    # and the relax actually recovers the true nadir orientations from real (noisy) matches
    err = max(qangle(got["orientation"][i], grid.orientation[i]) for i in range(grid.n_images))
    assert err < 5e-3, err
    # the plane found is the synthetic ground: z = 1e-3 x + 1e-2 y
    for x, y, z in got["plane"]:
        assert abs(z - (1e-3 * x + 1e-2 * y)) < 0.5
    g.close()
    ctx.close()
