"""Set-up of the ground-plane relax on the device (ochip_plane_setup_*, csrc/relax_setup.hip): the grid filter of
gridFilterMatchesPerImage (src/relax/relax_problem.cpp:234-309) and the residual-block list of
addRayTriangleMeasurementCost (:388-560) against the host's code on real linked graphs, bit for bit, and the ABI's
edge cases (ties and out-of-table matches are flagged, not decided; empty inputs)."""
import ctypes as C

import numpy as np
import pytest

from opencalibration_amd import capi, host, synth
from test_gpu_pipeline import perturbed

pytestmark = pytest.mark.gpu

EDGE = np.dtype([("cam_a", "<u4"), ("cam_b", "<u4"), ("model_a", "<u4"), ("model_b", "<u4"), ("n_inliers", "<u4"),
                 ("flags", "<u4"), ("inlier_offset", "<u8"), ("H", "<f8", 9)])
INLIER = np.dtype([("px1", "<f8", 2), ("px2", "<f8", 2), ("descriptor_score", "<f8")])


@pytest.fixture()
def checked():
    before = host.relax_setup_check(1)
    yield lambda: host.relax_setup_check(-1) - before
    host.relax_setup_check(0)


@pytest.mark.parametrize("cfg, seed", [("C1", 1), (dict(seed=5, rows=6, cols=12, feats=1024), 2)])
def test_device_blocks_equal_the_hosts_on_linked_graphs(checked, cfg, seed):
    grid = synth.make_grid(**(synth.CONFIGS[cfg] if isinstance(cfg, str) else cfg))
    ctx = capi.Context(0)
    g = host.Graph.from_synthetic(grid)
    g.link(ctx)
    start = perturbed(grid.orientation, 0.1, seed)
    g.set_orientations(start)
    got = g.relax_ground_plane(ctx, start)  # raises if a set-up's blocks differ from the host's
    assert checked() >= 1
    assert int(got["residual_blocks"]) > 0
    g.close()
    ctx.close()


def test_distorted_cameras_too(checked):
    """image_to_3d with lens distortion (the TinySolver inversion, csrc/undistort.hpp) runs inside both kernels."""
    grid = synth.make_grid(seed=9, rows=3, cols=4, feats=512, distortion=[-0.05, 0.01, 0.0, 1e-3, -1e-3])
    ctx = capi.Context(0)
    g = host.Graph.from_synthetic(grid)
    g.link(ctx)
    start = perturbed(grid.orientation, 0.05, 3)
    g.set_orientations(start)
    g.relax_ground_plane(ctx, start)
    assert checked() >= 1
    g.close()
    ctx.close()


def test_tied_scores_go_to_the_hosts_sort(checked):
    """Every inlier match twice: every cell's best score is shared by two matches, the device flags every edge and the
    host's walk over its std::sort decides which of the two stays; the blocks must still equal the host's own."""
    grid = synth.make_grid(seed=11, rows=3, cols=4, feats=512)
    ctx = capi.Context(0)
    g = host.Graph.from_synthetic(grid)
    g.link(ctx)
    g2 = host.Graph.from_synthetic(grid)
    for e in g.edges(with_distances=True):
        twice = lambda a: np.concatenate([a, a])
        g2.add_edge(e["source"], e["dest"], twice(e["px"]), twice(e["f1"]), twice(e["f2"]), twice(e["match_index"]), e["H"], e["dist"],
                    e["poses"], match_idx=e["match_idx"], is_homography=e["is_homography"])
    assert g2.num_edges == g.num_edges > 0
    start = perturbed(grid.orientation, 0.05, 5)
    g2.set_orientations(start)
    got = g2.relax_ground_plane(ctx, start)
    assert checked() >= 1 and int(got["residual_blocks"]) > 0
    g.close()
    g2.close()
    ctx.close()


def test_mesh_flavour_filters_on_the_device_too(checked):
    """setupGroundMeshProblem's gridFilterMatchesPerImage (grid fraction 0.1) through the same kernel: whitelists equal
    to the host's on every edge, then the usual mesh relax."""
    grid = synth.make_grid(seed=5, rows=4, cols=6, feats=1024)
    ctx = capi.Context(0)
    g = host.Graph.from_synthetic(grid)
    g.link(ctx)
    start = perturbed(grid.orientation, 0.05, 2)
    g.set_orientations(start)
    plane = g.relax(ctx, start, host.relax_options("ORIENTATION", "GROUND_PLANE"))
    seed = host.rebuild_mesh(grid.position, plane["surface"], minimal=True)
    n0 = checked()
    mesh = g.relax(ctx, plane["orientation"], host.relax_options("ORIENTATION", "GROUND_MESH"), 0.1, previous=seed)
    assert checked() > n0 and int(mesh["residual_blocks"]) > 0
    g.close()
    ctx.close()


def _setup(ctx, edges, inliers, cam_pos, cam_q, models, tri, frac=0.15):
    L = ctx.L
    vp = C.c_void_p
    L.ochip_plane_setup_create.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint64, vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_double,
                                           vp, vp, C.POINTER(vp)]
    L.ochip_plane_setup_blocks.argtypes = [vp, vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.ochip_plane_setup_destroy.argtypes = [vp]
    L.ochip_plane_setup_destroy.restype = None
    keep = np.zeros(max(len(inliers), 1), np.uint8)
    inexact = np.zeros(max(len(edges), 1), np.uint8)
    h = vp()
    ptr = lambda a: a.ctypes.data_as(vp)
    rc = L.ochip_plane_setup_create(ctx.h, ptr(edges), len(edges), ptr(inliers), len(inliers), ptr(cam_pos), ptr(cam_q), len(cam_pos),
                                    ptr(models), len(models), ptr(tri), frac, ptr(keep), ptr(inexact), C.byref(h))
    assert rc == 0, ctx.L.ochip_last_error(ctx.h).decode()
    n = C.c_uint64(0)
    assert L.ochip_plane_setup_blocks(h, None, None, None, 0, C.byref(n)) == 0
    a = np.zeros(max(n.value, 1), np.uint32)
    b = np.zeros(max(n.value, 1), np.uint32)
    rays = np.zeros((max(n.value, 1), 6))
    assert L.ochip_plane_setup_blocks(h, ptr(a), ptr(b), ptr(rays), n.value, C.byref(n)) == 0
    L.ochip_plane_setup_destroy(h)
    return keep[:len(inliers)], inexact[:len(edges)], a[:n.value], b[:n.value], rays[:n.value]


def _two_nadir_cameras():
    cam_pos = np.array([[0.0, 0.0, 100.0], [20.0, 0.0, 100.0]])
    cam_q = np.array([[1.0, 0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]])  # looking down (rotation by pi about x)
    models = np.array([[600.0, 400.0, 300.0, 0, 0, 0, 0, 0, 800.0, 600.0]])
    tri = np.array([-500.0, -500.0, 500.0, -500.0, 0.0, 500.0])
    return cam_pos, cam_q, models, tri


def _inliers_of_ground_points(pts, cam_pos, f=600.0, pp=(400.0, 300.0)):
    """Pixels of ground points (z = 0) in two cameras looking straight down (x right, y towards -Y)."""
    out = np.zeros(len(pts), INLIER)
    for k, key in enumerate(("px1", "px2")):
        d = pts - cam_pos[k, :2]
        out[key][:, 0] = pp[0] + f * d[:, 0] / 100.0
        out[key][:, 1] = pp[1] - f * d[:, 1] / 100.0
    out["descriptor_score"] = 0.9
    return out


def test_one_survivor_per_cell_and_ties_are_flagged():
    ctx = capi.Context(0)
    cam_pos, cam_q, models, tri = _two_nadir_cameras()
    rng = np.random.default_rng(4)
    pts = rng.uniform([-20, -30], [40, 30], size=(500, 2))
    inl = _inliers_of_ground_points(pts, cam_pos)
    inl["descriptor_score"] = rng.uniform(0.5, 1.0, len(inl))
    edges = np.zeros(1, EDGE)
    edges[0] = (0, 1, 0, 0, len(inl), 0, 0, np.eye(3).ravel())
    keep, inexact, a, b, rays = _setup(ctx, edges, inl, cam_pos, cam_q, models, tri)
    assert inexact[0] == 0
    for bit, key in ((1, "px1"), (2, "px2")):
        cells = np.floor(inl[key] / [800.0, 600.0] / 0.15).astype(int)
        kept = cells[(keep & bit) != 0]
        assert len(kept) == len({tuple(c) for c in kept})  # at most one per cell
        # every occupied cell of matches that score above zero keeps one: the rays all meet on the ground, so all do
        assert len({tuple(c) for c in kept}) == len({tuple(c) for c in cells})
    assert len(a) == int((keep != 0).sum()) and (a == 0).all() and (b == 1).all()
    assert np.allclose(np.linalg.norm(rays[:, :3], axis=1), 1) and np.allclose(np.linalg.norm(rays[:, 3:], axis=1), 1)
    # the same match twice: its cell's best score is shared - the caller's to decide
    best = int(np.flatnonzero(keep & 1)[0])
    twice = np.concatenate([inl, inl[best:best + 1]])
    edges["n_inliers"] = len(twice)
    assert _setup(ctx, edges, twice, cam_pos, cam_q, models, tri)[1][0] == 1
    # a match outside the image (cell index below zero): flagged as well
    out = np.concatenate([inl, _inliers_of_ground_points(np.array([[-68.3, 0.0]]), cam_pos)])  # pixel x = -10 and -130
    edges["n_inliers"] = len(out)
    assert _setup(ctx, edges, out, cam_pos, cam_q, models, tri)[1][0] == 1
    # a grid finer than the kernel's cell tables: every edge is the caller's
    edges["n_inliers"] = len(inl)
    assert _setup(ctx, edges, inl, cam_pos, cam_q, models, tri, frac=0.01)[1][0] == 1
    ctx.close()


def test_blocks_need_the_triangle_and_edges_keep_their_order():
    ctx = capi.Context(0)
    cam_pos, cam_q, models, tri = _two_nadir_cameras()
    rng = np.random.default_rng(6)
    pts = rng.uniform([-20, -30], [40, 30], size=(300, 2))
    inl = np.concatenate([_inliers_of_ground_points(pts[:200], cam_pos), _inliers_of_ground_points(pts[200:], cam_pos[::-1])])
    edges = np.zeros(3, EDGE)
    edges[0] = (0, 1, 0, 0, 200, 0, 0, np.eye(3).ravel())
    edges[1] = (1, 0, 0, 0, 0, 0, 200, np.eye(3).ravel())      # an edge without inliers
    edges[2] = (1, 0, 0, 0, 100, 0, 200, np.eye(3).ravel())
    keep, inexact, a, b, rays = _setup(ctx, edges, inl, cam_pos, cam_q, models, tri)
    assert not inexact.any()
    n0 = int((keep[:200] != 0).sum())
    assert (a[:n0] == 0).all() and (a[n0:] == 1).all() and len(a) == int((keep != 0).sum())
    far = np.array([5000.0, 5000.0, 6000.0, 5000.0, 5500.0, 6000.0])  # a triangle none of the points lie over
    assert len(_setup(ctx, edges, inl, cam_pos, cam_q, models, far)[2]) == 0
    # nothing at all
    none = _setup(ctx, np.zeros(0, EDGE), np.zeros(0, INLIER), cam_pos, cam_q, models, tri)
    assert len(none[2]) == 0
    ctx.close()


def test_bad_arguments_are_refused():
    ctx = capi.Context(0)
    cam_pos, cam_q, models, tri = _two_nadir_cameras()
    edges = np.zeros(1, EDGE)
    edges[0] = (0, 7, 0, 0, 0, 0, 0, np.eye(3).ravel())  # camera 7 of 2
    L = ctx.L
    vp = C.c_void_p
    L.ochip_plane_setup_create.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint64, vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_double,
                                           vp, vp, C.POINTER(vp)]
    h = vp()
    flag = np.zeros(1, np.uint8)
    ptr = lambda x: x.ctypes.data_as(vp)
    rc = L.ochip_plane_setup_create(ctx.h, ptr(edges), 1, None, 0, ptr(cam_pos), ptr(cam_q), 2, ptr(models), 1, ptr(tri), 0.15, None,
                                    ptr(flag), C.byref(h))
    assert rc == -1 and "out of range" in ctx.L.ochip_last_error(ctx.h).decode() and not h.value  # OCHIP_EINVAL
    ctx.close()
