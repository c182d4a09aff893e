"""INITIAL_PROCESSING as the reference pipelines it (Pipeline::Impl::initial_processing, src/pipeline/pipeline.cpp:522-570):
step k extracts batch k, links batch k - 1 against everything loaded before it and relaxes batch k - 2 as one group with two
rings of context cameras, the three stages' runners side by side, finalized in the reference's order
(och_initial_processing_*, csrc/host/initial_processing.cpp).  Checked: the stages side by side give the graph the stages
one after the other give (nothing a runner reads is written while it runs), and every relax stage lands where the oracle's
RelaxGroup restatement lands when it is fed the same graph state - the nodes up to one batch behind the relaxed batch's own,
the edges up to its own (the reference's state at that moment) - within 1e-6 rad, with the same number of solves."""
import numpy as np
import pytest

from opencalibration_amd import capi, host, pipeline, synth
from relax_fixtures import qangle

pytestmark = pytest.mark.gpu


def _edge_arrays(g):
    return [(e["source"], e["dest"], e["n_matches"], e["px"].tobytes(), e["f1"].tobytes(), e["match_index"].tobytes()) for e in g.edges()]


def test_side_by_side_equals_one_after_the_other():
    ctx = capi.Context(0)
    grid = synth.make_grid(4, 6, feats=64, seed=9)
    images, shape = pipeline.synthetic_views(ctx, grid, seed=3)
    batch = 5                                              # (24 views: a last batch of 4)
    ga, sa = pipeline.run_initial_processing(ctx, grid, images, shape, batch=batch)
    gb, sb = pipeline.run_initial_processing(ctx, grid, images, shape, batch=batch, sequential=True)
    assert sa["batches"] == sb["batches"] == 5 and sa["steps"] == sb["steps"] == 7      # two more steps drain the link and relax stages
    assert ga.node_ids == gb.node_ids and ga.num_edges == gb.num_edges > 100
    assert _edge_arrays(ga) == _edge_arrays(gb)
    assert np.array_equal(ga.orientations(), gb.orientations())
    assert sa["solves"] == sb["solves"] and sa["lm_iterations"] == sb["lm_iterations"]
    err = pipeline.orientation_errors(ga.orientations(), grid.orientation)
    assert np.all(np.isfinite(err)) and np.median(err) < 5e-3
    ga.close(), gb.close()
    ctx.synth_views_free(images)
    ctx.close()


def test_every_relax_stage_matches_the_oracle_on_the_references_graph_state(oracle):
    ctx = capi.Context(0)
    grid = synth.make_grid(4, 6, feats=64, seed=9)
    images, shape = pipeline.synthetic_views(ctx, grid, seed=3)
    n, h, w = shape
    batch = 6
    opts_o = oracle.options("ORIENTATION", "GROUND_PLANE")
    g = host.Graph()
    mid = g.add_model(grid.model)
    ip = g.initial_processing(ctx)
    rx = oracle.RxGraph()
    rx.add_model(grid.model, 1)
    index_of, in_rx, edges_in_rx = {}, 0, 0
    batches = [list(range(lo, min(lo + batch, n))) for lo in range(0, n, batch)]
    step = 0
    compared = 0
    while step < len(batches) or ip.pending:
        # what this step's relax stage sees is the graph the steps before left: the oracle relaxes batch step - 2 on its copy
        exp = None
        if step >= 2 and step - 2 < len(batches):
            knn = oracle.knn10_bruteforce(grid.position[:in_rx, :2])
            exp = rx.relax_group(batches[step - 2], knn, 2, opts_o)
        if step < len(batches):
            idx = batches[step]
            st = ip.step(images + idx[0] * h * w * 3, mid, grid.position[idx[0]:idx[-1] + 1], device_shape=(len(idx), h, w))
        else:
            st = ip.step()
        if exp is not None:
            assert int(st["relax_solves"]) == exp["solves"] > len(batches[step - 2])
            eo, go = rx.orientations(), g.orientations()
            upto = batches[step - 2][-1] + 1
            assert np.all(np.isfinite(go[:upto]))
            worst = max(qangle(eo[i], go[i]) for i in range(upto))
            assert worst < 1e-6, (step, worst)
            compared += 1
        # the step's finalize: the loaded batch's nodes and the linked batch's edges enter the oracle's copy as well
        while in_rx < g.num_nodes:
            p = g.node_payload(in_rx)
            assert np.all(np.isnan(p["orientation"]))        # images arrive without an orientation
            index_of[g.node_ids[in_rx]] = in_rx
            rx.add_node(grid.position[in_rx], np.full(4, np.nan), 0, p["loc"], p["path"])
            in_rx += 1
        for ed in g.edges(with_distances=True)[edges_in_rx:]:
            rx.add_edge(index_of[ed["source"]], index_of[ed["dest"]], ed["px"], ed["f1"], ed["f2"], ed["match_index"], ed["H"], ed["dist"])
            edges_in_rx += 1
        step += 1
    assert compared == len(batches) and step == len(batches) + 2
    err = pipeline.orientation_errors(g.orientations(), grid.orientation)
    assert np.median(err) < 5e-3
    ip.close()
    g.close()
    ctx.synth_views_free(images)
    ctx.close()
