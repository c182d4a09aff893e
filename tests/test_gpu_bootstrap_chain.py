"""The NaN bootstrap of runGroundPlane (src/relax/relax.cpp:52-80) as one resident launch (csrc/relax_chain.hip,
ochip_plane_chain_*): INITIAL_PROCESSING's schedule - batches that arrive without orientations, one relax group per batch
with two rings of context (pipeline.cpp:545-546, relax_stage.cpp:95) - through three routes of the same library:
the resident launch (default), the same code one phase per launch (OCHIP_TEST_HOOKS=chain_stepped: bit for bit), and round 5's
host loop (OCHIP_TEST_HOOKS=host_bootstrap: a problem and two solves per camera on the pair-record engine - the same
solves and iterations, orientations to 1e-7: the two sum the normal equations in different orders and stop at the same
iteration).  The oracle comparison of the default route is tests/test_gpu_incremental.py."""
import numpy as np
import pytest

from opencalibration_amd import capi, host, synth
from relax_fixtures import qangle

pytestmark = pytest.mark.gpu


def _schedule(ctx, grid, batch, hook, monkeypatch):
    """Load + link + relax batch after batch; returns the orientations after every batch and the stages' figures."""
    if hook:
        monkeypatch.setenv("OCHIP_TEST_HOOKS", hook)
    else:
        monkeypatch.delenv("OCHIP_TEST_HOOKS", raising=False)
    opts = host.relax_options("ORIENTATION", "GROUND_PLANE")
    g = host.Graph()
    m = g.add_model(grid.model)
    n = grid.n_images
    out, stats = [], []
    for lo in range(0, n, batch):
        idx = list(range(lo, min(lo + batch, n)))
        for i in idx:
            loc, st, de, _ = grid.image(i)
            g.add_image(loc, st, de, grid.num_sparse[i], m, grid.position[i])
        ori = g.orientations().copy()
        ori[lo:] = np.nan
        g.set_orientations(ori)
        g.link(ctx, node_ids=[g.node_ids[i] for i in idx])
        st = g.relax_stage(ctx, opts, node_ids=[g.node_ids[i] for i in idx], disable_parallelism=True)
        out.append(g.orientations().copy())
        stats.append((int(st["solves"]), int(st["iterations_total"]), int(st["residual_blocks"])))
    g.close()
    monkeypatch.delenv("OCHIP_TEST_HOOKS", raising=False)
    return out, stats


@pytest.mark.parametrize("rows,cols,batch", [(3, 8, 8), (10, 20, 20)])
def test_resident_launch_equals_its_stepped_form_and_the_host_loop(rows, cols, batch, monkeypatch):
    ctx = capi.Context(0)
    grid = synth.make_grid(rows, cols, feats=512, seed=11)
    resident, s_res = _schedule(ctx, grid, batch, None, monkeypatch)
    stepped, s_step = _schedule(ctx, grid, batch, "chain_stepped", monkeypatch)
    loop, s_loop = _schedule(ctx, grid, batch, "host_bootstrap", monkeypatch)
    n_batches = len(resident)
    assert n_batches == (rows * cols + batch - 1) // batch
    for b in range(n_batches):
        upto = min((b + 1) * batch, grid.n_images)
        assert np.all(np.isfinite(resident[b][:upto])), b
        # the same code with a launch per phase: bit for bit
        assert np.array_equal(resident[b], stepped[b], equal_nan=True), b
        assert s_res[b] == s_step[b], (b, s_res[b], s_step[b])
        # round 5's host loop: the same solves and LM iterations, the same orientations up to rounding
        assert s_res[b][0] == s_loop[b][0] and s_res[b][2] == s_loop[b][2], (b, s_res[b], s_loop[b])
        assert s_res[b][1] == s_loop[b][1], (b, s_res[b], s_loop[b])
        worst = max(qangle(resident[b][i], loop[b][i]) for i in range(upto))
        assert worst < 1e-7, (b, worst)
    # the later batches of the larger grid are relaxed one camera at a time against the graph (relax.cpp:61-68): both modes ran
    err = np.array([qangle(resident[-1][i], grid.orientation[i]) for i in range(grid.n_images)])
    assert np.median(err) < 2e-3
    ctx.close()


def test_a_chain_that_stops_half_way_hands_over_to_the_host_loop(monkeypatch):
    """The chain gives up when a step needs the host's grid filter (a tie for a cell's best score) or a wait on the device runs
    into its limit: the poses it finished are written back and the host loop continues from the next one.
    OCHIP_TEST_HOOKS=chain_partial makes it stop after half of every group's cameras: the same solves, iterations and (to
    rounding) orientations as the chain alone and the host loop alone."""
    ctx = capi.Context(0)
    grid = synth.make_grid(4, 10, feats=512, seed=4)
    whole, s_whole = _schedule(ctx, grid, 10, None, monkeypatch)
    half, s_half = _schedule(ctx, grid, 10, "chain_partial", monkeypatch)
    loop, s_loop = _schedule(ctx, grid, 10, "host_bootstrap", monkeypatch)
    for b in range(len(whole)):
        upto = min((b + 1) * 10, grid.n_images)
        assert s_half[b] == s_whole[b] == s_loop[b], (b, s_half[b], s_whole[b], s_loop[b])
        assert max(qangle(half[b][i], whole[b][i]) for i in range(upto)) < 1e-7
        assert max(qangle(half[b][i], loop[b][i]) for i in range(upto)) < 1e-7
    ctx.close()
