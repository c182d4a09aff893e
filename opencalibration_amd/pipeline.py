"""End-to-end driver of the hot path on one GPU: images -> extract -> link (match + RANSAC) -> relax.

Mirrors what Pipeline::Impl::initial_processing strings together (src/pipeline/pipeline.cpp:522-570): the
load stage's extract_features per image, the link stage over kNN(10) pairs, one relax group over the linked
cameras.  Used by bench.py and the end-to-end tests; the images come from the synthetic view renderer
(data only) and are resident in HBM before any timed region starts.
"""
import os
import time

import numpy as np

from . import host, synth


def synthetic_views(ctx, grid, seed=7, block=None):
    """Render one 4000x3000-class view per camera of `grid` into HBM.  Returns (device pointer, (n, h, w)).
    block = (first, count): only those cameras' views (one survey over several ranks: a rank holds its block)."""
    w, h = int(grid.model[8]), int(grid.model[9])
    f, pp = float(grid.model[0]), (float(grid.model[1]), float(grid.model[2]))
    # blob lattice: ~21 px apart in the 1600-px working image, i.e. (max(w,h)/1600) * 21 full-resolution pixels
    gsd = (grid.position[0, 2] - (grid.plane[0] * grid.position[0, 0] + grid.plane[1] * grid.position[0, 1])) / f
    spacing = float(os.environ.get("OCHIP_BLOB_SPACING", "21.0")) * (max(w, h) / 1600.0) * gsd
    origin = (float(grid.position[:, 0].min() - 500.0), float(grid.position[:, 1].min() - 500.0))
    lo, cnt = (0, grid.n_images) if block is None else block
    ptr = ctx.synth_views(grid.position[lo:lo + cnt], grid.orientation[lo:lo + cnt], w, h, f, pp, grid.plane, spacing, origin,
                          seed=seed)
    return ptr, (cnt, h, w)


def relax_step(ctx, g, start_orientation, res, t):
    """The relax stage of run() on its own (the bench pipelines it with the next survey's load + link)."""
    t0, c0 = time.perf_counter(), time.process_time()
    rel = g.relax_ground_plane(ctx, start_orientation)
    ctx.synchronize()
    t["relax"] = time.perf_counter() - t0
    t["host_cpu_relax"] = time.process_time() - c0
    res["relax"] = rel
    return rel


def run(ctx, grid, images_ptr, shape, start_orientation, max_keypoints=30000, overlap=True, relax=True, host_images=None):
    """load (extract) -> link -> relax.  Returns (graph, result dict, stage seconds).  overlap: the load and link
    stages run overlapped (och_graph_load_link_images) as the reference's pipeline overlaps the stages of consecutive
    batches; the graph is the same either way.  relax=False: stop after the link stage (relax_step() does the rest).
    host_images: an (n, h, w, 3) uint8 array in host memory to start from instead of the views in HBM (the upload is then
    part of the load stage)."""
    n, h, w = shape
    t = {}
    g = host.Graph()
    mid = g.add_model(grid.model)
    if overlap:
        t0, c0 = time.perf_counter(), time.process_time()
        feats_mean, sparse_mean, link_timers, (t_ex, t_all) = g.load_link_images(
            ctx, images_ptr if host_images is None else host_images, mid, grid.position, start_orientation, max_keypoints,
            device_shape=(n, h, w) if host_images is None else None)
        t["extract"], t["link"] = t_ex, t_all - t_ex   # link = what the linking adds after the last features (native clocks:
        #                                                  the wait for the previous survey's extraction is in neither)
        t["host_cpu_load_link"] = time.process_time() - c0                 # CPU seconds of all host threads
        if not relax:
            return g, dict(features_per_image=feats_mean, sparse_per_image=sparse_mean, link_timers=link_timers, edges=g.num_edges), t
        t0, c0 = time.perf_counter(), time.process_time()
        rel = g.relax_ground_plane(ctx, start_orientation)
        ctx.synchronize()
        t["relax"] = time.perf_counter() - t0
        t["host_cpu_relax"] = time.process_time() - c0
        res = dict(features_per_image=feats_mean, sparse_per_image=sparse_mean, link_timers=link_timers, relax=rel,
                   edges=g.num_edges)
        return g, res, t
    t0, c0 = time.perf_counter(), time.process_time()
    feats_mean, sparse_mean = g.load_images(ctx, images_ptr, mid, grid.position, max_keypoints, device_shape=(n, h, w))
    t["extract"] = time.perf_counter() - t0
    t["host_cpu_extract"] = time.process_time() - c0   # CPU seconds of all host threads
    g.set_orientations(start_orientation)
    t0, c0 = time.perf_counter(), time.process_time()
    link_timers = g.link(ctx)
    t["link"] = time.perf_counter() - t0
    t["host_cpu_link"] = time.process_time() - c0
    t0, c0 = time.perf_counter(), time.process_time()
    rel = g.relax_ground_plane(ctx, start_orientation)
    ctx.synchronize()
    t["relax"] = time.perf_counter() - t0
    t["host_cpu_relax"] = time.process_time() - c0
    res = dict(features_per_image=feats_mean, sparse_per_image=sparse_mean, link_timers=link_timers, relax=rel,
               edges=g.num_edges)
    return g, res, t


def run_incremental(ctx, grid, images_ptr, shape, batch=100, max_keypoints=30000):
    """The reference's own schedule of INITIAL_PROCESSING (Pipeline::Impl::initial_processing, src/pipeline/pipeline.cpp:522-570):
    the survey arrives in batches; every batch is loaded (extract) and linked against everything loaded so far, and its
    cameras - which start WITHOUT an orientation (NaN) - are relaxed as one group {ORIENTATION, GROUND_PLANE} together with two
    rings of already oriented context cameras held fixed (RelaxStage::init with the batch's ids, relax_group.cpp:40-66);
    relax() initialises the NaN cameras one at a time, a solve each, before the group's solve (src/relax/relax.cpp:52-80).
    Returns (graph, dict of counts and seconds)."""
    n, h, w = shape
    g = host.Graph()
    mid = g.add_model(grid.model)
    opts = host.relax_options("ORIENTATION", "GROUND_PLANE")
    out = dict(batches=0, solves=0, lm_iterations=0, residual_blocks=0, load_link_s=0.0, relax_s=0.0)
    t_all = time.perf_counter()
    for lo in range(0, n, batch):
        cnt = min(batch, n - lo)
        t0 = time.perf_counter()
        nan = np.full((cnt, 4), np.nan)
        g.load_link_images(ctx, images_ptr + lo * h * w * 3, mid, grid.position[lo:lo + cnt], nan, max_keypoints, device_shape=(cnt, h, w))
        out["load_link_s"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        st = g.relax_stage(ctx, opts, node_ids=g.node_ids[lo:lo + cnt], disable_parallelism=True)   # (pipeline.cpp:545-546)
        ctx.synchronize()
        out["relax_s"] += time.perf_counter() - t0
        out["batches"] += 1
        out["solves"] += int(st["solves"])
        out["lm_iterations"] += int(st["iterations_total"])
        out["residual_blocks"] += int(st["residual_blocks"])
    out["seconds"] = time.perf_counter() - t_all
    out["edges"] = g.num_edges
    return g, out


def run_initial_processing(ctx, grid, images_ptr, shape, batch=100, max_keypoints=30000, sequential=False):
    """INITIAL_PROCESSING exactly as the reference schedules it (Pipeline::Impl::initial_processing, src/pipeline/pipeline.cpp:522-570;
    host.InitialProcessing): the survey arrives in batches WITHOUT orientations; step k extracts batch k, links batch k - 1
    against everything loaded before it and relaxes batch k - 2 as one group {ORIENTATION, GROUND_PLANE} with two rings of
    context cameras - the three side by side - until the pipeline is drained.  Returns (graph, dict of counts and seconds)."""
    n, h, w = shape
    g = host.Graph()
    mid = g.add_model(grid.model)
    ip = g.initial_processing(ctx)
    out = dict(batches=0, steps=0, solves=0, lm_iterations=0, load_runner_s=0.0, link_runner_s=0.0, relax_runner_s=0.0,
               relax_device_s=0.0, relax_setup_host_s=0.0, step_seconds=[])
    t_all = time.perf_counter()
    lo = 0
    while lo < n or ip.pending:
        if lo < n:
            cnt = min(batch, n - lo)
            st = ip.step(images_ptr + lo * h * w * 3, mid, grid.position[lo:lo + cnt], max_keypoints, device_shape=(cnt, h, w),
                         sequential=sequential)
            lo += cnt
            out["batches"] += 1
        else:
            st = ip.step(sequential=sequential)
        out["steps"] += 1
        out["solves"] += int(st["relax_solves"])
        out["lm_iterations"] += int(st["relax_iterations"])
        for k in ("load_runner_s", "link_runner_s", "relax_runner_s", "relax_device_s", "relax_setup_host_s"):
            out[k] += st[k]
        out["step_seconds"].append(round(st["step_s"], 4))
        out.setdefault("step_stage_seconds", []).append([round(st[k], 4) for k in ("init_s", "load_runner_s", "link_runner_s", "relax_runner_s",
                                                                                   "finalize_s")])
    ctx.synchronize()
    out["seconds"] = time.perf_counter() - t_all
    out["edges"] = g.num_edges
    ip.close()
    return g, out


def perturbed_orientations(grid, sigma=0.1, seed=99):
    """True orientation with a `sigma` rad error about a random axis (test/test_relax.cpp:421)."""
    rng = np.random.default_rng(seed)
    axes = rng.normal(size=(grid.n_images, 3))
    axes /= np.linalg.norm(axes, axis=1, keepdims=True)
    dq = np.concatenate([axes * np.sin(sigma / 2), np.full((grid.n_images, 1), np.cos(sigma / 2))], axis=1)
    return synth.quat_mul(grid.orientation, dq)


def orientation_errors(est, truth):
    dots = np.abs(np.sum(est * truth, axis=1) / (np.linalg.norm(est, axis=1) * np.linalg.norm(truth, axis=1)))
    return 2 * np.arccos(np.clip(dots, 0, 1))
