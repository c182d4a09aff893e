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


def _edge_signature(g):
    out = []
    for e in g.edges():
        out.append((e["source"], e["dest"], e["n_matches"], e["n_inliers"], e["H"].tobytes(), e["f1"].tobytes(),
                    e["f2"].tobytes(), e["poses"].tobytes()))
    return out


def test_link_runners_do_not_change_the_graph(monkeypatch):
    """The link stage splits its pairs over concurrent batch runners (sibling device contexts); finalize restores
    the reference's deterministic edge order (link_stage.cpp:123-127), so 1 and 3 runners give the same graph."""
    grid = synth.make_grid(seed=5, rows=10, cols=20, feats=512)
    ctx = capi.Context(0)
    sigs = []
    for runners in ("1", "3"):
        monkeypatch.setenv("OCHIP_LINK_RUNNERS", runners)
        g = host.Graph.from_synthetic(grid)
        g.link(ctx)
        sigs.append(_edge_signature(g))
        g.close()
    assert len(sigs[0]) > 1000 and sigs[0] == sigs[1]
    ctx.close()


def test_images_to_orientations(monkeypatch, oracle):
    """The whole path from pixels: rendered views -> load stage (extract on the device, chunked over two device
    contexts) -> link -> relax.  Chunking must not change a feature; the oracle's extract_features of a view
    fetched back from HBM is what the graph holds (checked through the matches it produces); the relax recovers
    the nadir orientations."""
    from opencalibration_amd import pipeline

    grid = synth.make_grid(seed=3, rows=2, cols=3, feats=64)
    ctx = capi.Context(0)
    images, shape = pipeline.synthetic_views(ctx, grid, seed=11)
    n, h, w = shape
    start = pipeline.perturbed_orientations(grid, 0.1, 4)
    sigs, results = [], []
    # the last variant overlaps the load and link stages (ranges of links start while later chunks are extracted)
    for chunk, streams, overlap in (("50", "1", False), ("2", "2", False), ("2", "2", True), ("1", "3", True)):
        monkeypatch.setenv("OCHIP_EXTRACT_CHUNK", chunk)
        monkeypatch.setenv("OCHIP_EXTRACT_STREAMS", streams)
        g, res, _ = pipeline.run(ctx, grid, images, shape, start, overlap=overlap)
        sigs.append(_edge_signature(g))
        results.append(res)
        g.close()
    assert all(s == sigs[0] for s in sigs[1:]) and len(sigs[0]) >= 2 * (n - 1)
    # extract parity on a real 4000 x 3000 view: device load stage == oracle extract_features
    feats = host.extract_features_batch(ctx, images, 30000, device_shape=(1, h, w))[0]
    view = ctx.synth_views_read(images, 0, w, h)
    eloc, est, edesc, ens = oracle.extract_features(view)
    assert feats[3] == ens and np.array_equal(feats[0], eloc) and np.array_equal(feats[1], est)
    assert np.array_equal(feats[2], edesc) and len(est) > 5000
    err = pipeline.orientation_errors(results[0]["relax"]["orientation"], grid.orientation)
    assert np.max(err) < 5e-3, err
    ctx.synth_views_free(images)
    ctx.close()


def test_second_batch_links_against_the_first(monkeypatch):
    """Incremental loading, as the reference's pipeline does it batch by batch: the overlapped load + link of a SECOND
    batch must link the new images against the ones already in the graph (their subsets and rays are prepared although
    no range waits for them) - the same graph as load_images + link over the same two batches."""
    from opencalibration_amd import pipeline

    grid = synth.make_grid(seed=3, rows=2, cols=4, feats=64)
    ctx = capi.Context(0)
    images, (n, h, w) = pipeline.synthetic_views(ctx, grid, seed=11)
    half, bytes_per_image = n // 2, h * w * 3
    start = pipeline.perturbed_orientations(grid, 0.1, 4)

    def batches(g, overlapped):
        mid = g.add_model(grid.model)
        for lo, hi in ((0, half), (half, n)):
            ptr = images + lo * bytes_per_image
            if overlapped:
                g.load_link_images(ctx, ptr, mid, grid.position[lo:hi], start[lo:hi], 30000, device_shape=(hi - lo, h, w))
            else:
                g.load_images(ctx, ptr, mid, grid.position[lo:hi], 30000, device_shape=(hi - lo, h, w))
                g.set_orientations(start[:hi])
                g.link(ctx, node_ids=g.node_ids[lo:hi])
        return _edge_signature(g)

    monkeypatch.setenv("OCHIP_EXTRACT_CHUNK", "2")
    a, b = host.Graph(), host.Graph()
    sa, sb = batches(a, True), batches(b, False)
    assert a.node_ids == b.node_ids and sa == sb
    first = set(a.node_ids[:half])
    crossing = [s for s in sa if (s[0] in first) != (s[1] in first)]
    assert len(crossing) >= half and all(s[3] > 6 for s in crossing)        # edges between the batches, with inliers
    a.close(), b.close()
    ctx.synth_views_free(images)
    ctx.close()


def test_device_tails_equal_the_host_tails(monkeypatch):
    """Round 3 moved the tail of extract_features (std::sort by response, suppression, records) and the tail of
    match_features_subset (ratio test, std::sort by distance) with PROSAC's order and the 40 px subsets to the device.  The round-2
    host routes are still there (OCHIP_TEST_HOOKS=host_tail,host_sort,host_subset): from the same pixels both must give the same feature lists
    and the same graph."""
    from opencalibration_amd import pipeline

    grid = synth.make_grid(seed=5, rows=2, cols=3, feats=64)
    ctx = capi.Context(0)
    images, shape = pipeline.synthetic_views(ctx, grid, seed=13)
    n, h, w = shape
    start = pipeline.perturbed_orientations(grid, 0.1, 4)
    feats, sigs = {}, {}
    for mode in ("device", "host"):
        if mode == "host":
            monkeypatch.setenv("OCHIP_TEST_HOOKS", "host_tail,host_sort,host_subset")
        feats[mode] = host.extract_features_batch(ctx, images, 30000, device_shape=(n, h, w))
        g, _, _ = pipeline.run(ctx, grid, images, shape, start, relax=False)
        sigs[mode] = _edge_signature(g)
        g.close()
    for a, b in zip(feats["device"], feats["host"]):
        assert a[3] == b[3] and len(a[1]) > 5000
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert sigs["device"] == sigs["host"] and len(sigs["device"]) >= 2 * (n - 1)
    ctx.synth_views_free(images)
    ctx.close()
