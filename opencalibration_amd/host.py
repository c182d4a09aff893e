"""ctypes binding of liboc_host.so (include/oc_host.h): the C++ host side of the hot path."""
import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboc_host.so")

_lib = None
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


# ochip_relax_exchange_fn (include/ochip.h): all-gather the per-pair record arrays of a sharded relax in place
RELAX_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64)


def effective_cpus():
    """Host cores this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU
    boxes expose 256 hardware threads but run the job under a 16-CPU cfs quota; 256 OpenMP threads inside
    such a quota only thrash)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def host_threads():
    """OpenMP team size for the host phases.  They are short bursts between device phases (tens of ms of every
    100 ms cfs period), so a team larger than the quota finishes them sooner without exhausting the period's
    budget - provided idle members sleep (OMP_WAIT_POLICY=passive, set in load()) and threads waiting for the device
    block (OCHIP_BLOCKING_SYNC, default on).  Measured on the 16-CPU-quota MI355X boxes, C3 ms per step:
    24 threads 624, 32 600, 48 586, 64 563, 96 568, 128 582 (with spinning waits 32 threads took 668 and 64 were
    throttled)."""
    return max(1, min(len(os.sched_getaffinity(0)), 4 * effective_cpus()))


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise capi.OchipError(f"{LIB_PATH} is missing: run `python -m opencalibration_amd.build`")
        os.environ.setdefault("OMP_NUM_THREADS", str(host_threads()))  # read by libgomp when the library loads
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")           # idle team members sleep (CPU quota, see bench.py)
        capi.load()  # libochip.so first (liboc_host.so links against it)
        L = C.CDLL(LIB_PATH)
        L.och_subsample.restype = C.c_size_t
        L.och_subsample.argtypes = [_f64p, _f32p, C.c_size_t, C.c_double, C.c_size_t, _u64p]
        L.och_matches_from_device.restype = C.c_size_t
        L.och_matches_from_device.argtypes = [C.c_void_p, _u64p, C.c_size_t, _u64p, C.c_size_t, _u64p, _u64p, _f64p]
        vp, u32, u64, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t
        L.och_graph_create.restype = vp
        L.och_graph_destroy.argtypes = [vp]
        L.och_graph_destroy.restype = None
        L.och_last_error.argtypes = [vp]
        L.och_last_error.restype = C.c_char_p
        L.och_graph_add_model.argtypes = [vp, _f64p]
        L.och_graph_add_model.restype = u32
        L.och_graph_add_image.argtypes = [vp, _f64p, _f32p, _u64p, sz, sz, u32, _f64p]
        L.och_graph_add_image.restype = u64
        L.och_graph_add_edge.restype = u64
        L.och_graph_add_edge.argtypes = [vp, u64, u64, vp, C.c_int, sz, _f64p, _u64p, sz, vp, vp, vp]
        L.och_link_match_work.argtypes = [vp, _f64p]
        L.och_link_match_work.restype = None
        L.och_graph_get_orientations.argtypes = [vp, _f64p]
        L.och_graph_get_orientations.restype = None
        L.och_graph_num_nodes.argtypes = [vp]
        L.och_graph_num_nodes.restype = sz
        L.och_graph_num_edges.argtypes = [vp]
        L.och_graph_num_edges.restype = sz
        L.och_graph_node_ids.argtypes = [vp, _u64p]
        L.och_link_stage_run.argtypes = [vp, vp, _u64p, sz, C.c_int, _f64p]
        L.och_link_debug_count.argtypes = [vp]
        L.och_link_debug_count.restype = sz
        L.och_link_debug_pair.argtypes = [vp, sz, _u64p, _u64p, _f64p, np.ctypeslib.ndpointer(np.uint32)]
        L.och_link_debug_matches.argtypes = [vp, sz, _u64p, _u64p, _f64p, np.ctypeslib.ndpointer(np.uint8)]
        L.och_graph_edge_info.argtypes = [vp, sz, _u64p, _u64p, _f64p, _f64p]
        L.och_graph_edge_inliers.argtypes = [vp, sz, _u64p, _u64p, _u64p, _f64p]
        L.och_graph_edge_match_distances.argtypes = [vp, sz, _f64p]
        L.och_graph_edge_matches.argtypes = [vp, sz, vp, vp, C.POINTER(C.c_int)]
        L.och_graph_edge_matches.restype = None
        L.och_graph_set_orientations.argtypes = [vp, _f64p]
        u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
        L.och_relax_ground_plane.argtypes = [vp, sz, _f64p, _f64p, _f64p, sz, _u64p, _f64p, sz, _u64p, _u64p, _f64p, u8p,
                                             _u64p, _f64p, _u64p, vp, vp, sz, _u64p, _f64p, _f64p]
        L.och_relax_last_error.restype = C.c_char_p
        L.och_debug_relax_setup_check.argtypes = [C.c_int]
        L.och_graph_relax_ground_plane.argtypes = [vp, vp, _f64p, _f64p, _f64p]
        L.och_graph_relax_ground_plane_sharded.argtypes = [vp, vp, _f64p, _f64p, _f64p, u32, u32, vp, vp]
        L.och_surface_create.restype = vp
        L.och_surface_destroy.argtypes = [vp]
        L.och_surface_destroy.restype = None
        L.och_surface_counts.argtypes = [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
        L.och_surface_counts.restype = None
        L.och_surface_get.argtypes = [vp, vp, vp, vp]
        L.och_surface_get.restype = None
        L.och_surface_set.argtypes = [vp, sz, _f64p, sz, _u64p, sz, _f64p]
        L.och_surface_set.restype = None
        L.och_surface_set_heights.argtypes = [vp, _f64p]
        L.och_surface_set_heights.restype = None
        L.och_rebuild_mesh.argtypes = [_f64p, sz, vp, C.c_int, vp]
        L.och_rebuild_mesh.restype = None
        L.och_relax.argtypes = [vp, sz, _f64p, _f64p, _f64p, _u64p, _f64p, sz, _u64p, _f64p, sz, _u64p, _u64p, vp, u8p, _u64p,
                                _f64p, _u64p, _u64p, vp, vp, sz, _u64p, u32, C.c_double, vp, vp, _f64p, vp]
        L.och_relax_ex.argtypes = L.och_relax.argtypes + [vp, C.c_int, vp, vp, sz, C.POINTER(sz)]
        L.och_graph_relax.argtypes = [vp, vp, _f64p, u32, C.c_double, vp, vp, _f64p]
        L.och_graph_relax_sharded.argtypes = [vp, vp, _f64p, u32, C.c_double, vp, vp, _f64p, u32, u32, vp, vp]
        i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
        L.och_relax_stage_run.argtypes = [vp, vp, vp, sz, C.c_int, C.c_int, u32, C.c_double, sz, vp, vp, i64p, _f64p]
        L.och_relax_stage_begin.restype = vp
        L.och_relax_stage_begin.argtypes = [vp, vp, sz, C.c_int, C.c_int, u32, C.c_double, sz, vp, i64p]
        L.och_relax_stage_num_groups.argtypes = [vp]
        L.och_relax_stage_num_groups.restype = sz
        L.och_relax_stage_run_groups.argtypes = [vp, vp, u32, u32]
        L.och_relax_stage_export.argtypes = [vp, u32, u32, C.POINTER(vp), C.POINTER(u64)]
        L.och_relax_stage_import.argtypes = [vp, vp, u64]
        L.och_relax_stage_end.argtypes = [vp, vp, _f64p]
        L.och_relax_partition.restype = sz
        L.och_relax_partition.argtypes = [vp, sz, i64p, i64p]
        L.och_merge_surfaces.argtypes = [C.POINTER(vp), sz, vp]
        L.och_merge_surfaces.restype = None
        L.och_homography_decompose.argtypes = [_f64p, _f64p, sz, _f64p]
        L.och_image_to_3d.argtypes = [_f64p, sz, _f64p, _f64p]
        L.och_ransac_epipolar.restype = C.c_double
        L.och_ransac_epipolar.argtypes = [vp, C.c_int, _f64p, vp, sz, C.c_double, _f64p, u8p, np.ctypeslib.ndpointer(np.uint32)]
        L.och_extract_tail_prepared.restype = C.c_size_t
        L.och_extract_tail_prepared.argtypes = [C.c_void_p, _f32p, C.c_void_p, u32, C.c_int, u32, C.c_double, _f64p, _f32p, _u64p, _u64p]
        L.och_extract_tail.restype = C.c_size_t
        L.och_extract_tail.argtypes = [_f32p, _u64p, u32, C.c_double, _f64p, _f32p, _u64p, _u64p]
        L.och_graph_set_model.argtypes = [vp, u32, _f64p]
        L.och_graph_refit_edges.argtypes = [vp, vp]
        L.och_extract_features_batch.argtypes = [vp, vp, u32, C.c_int, C.c_int, u32, u32, vp, vp, vp, vp, vp, C.c_int]
        L.och_extract_last_error.restype = C.c_char_p
        L.och_graph_load_images.argtypes = [vp, vp, vp, u32, C.c_int, C.c_int, u32, C.c_int, u32, _f64p, _u64p, _f64p]
        L.och_graph_load_link_images.argtypes = [vp, vp, vp, u32, C.c_int, C.c_int, u32, C.c_int, u32, _f64p, vp, _u64p,
                                                 _f64p, _f64p, _f64p]
        L.och_initial_processing_create.restype = vp
        L.och_initial_processing_create.argtypes = [vp, vp]
        L.och_initial_processing_destroy.restype = None
        L.och_initial_processing_destroy.argtypes = [vp]
        L.och_initial_processing_pending.argtypes = [vp]
        L.och_initial_processing_step.argtypes = [vp, vp, u32, C.c_int, C.c_int, u32, C.c_int, u32, vp, C.c_int, vp, _f64p]
        L.och_graph_to_json.argtypes = [vp, C.POINTER(sz)]
        L.och_graph_to_json.restype = vp
        L.och_free.argtypes = [vp]
        L.och_free.restype = None
        L.och_graph_from_json.argtypes = [vp, C.c_char_p, sz]
        L.och_graph_save_json.argtypes = [vp, C.c_char_p]
        L.och_graph_load_json.argtypes = [vp, C.c_char_p]
        L.och_graph_node_table.argtypes = [vp, vp, vp, vp, vp]
        L.och_graph_node_table.restype = None
        L.och_graph_node_payload.argtypes = [vp, sz, vp, vp, vp, vp, vp]
        L.och_graph_node_path.argtypes = [vp, sz]
        L.och_graph_node_path.restype = C.c_char_p
        L.och_graph_set_node_path.argtypes = [vp, sz, C.c_char_p]
        L.och_graph_num_models.argtypes = [vp]
        L.och_graph_num_models.restype = sz
        L.och_graph_get_model.argtypes = [vp, u32, _f64p]
        L.och_surface_save_ply.argtypes = [vp, C.c_char_p]
        L.och_surface_load_ply.argtypes = [vp, C.c_char_p]
        L.och_surface_num_clouds.argtypes = [vp]
        L.och_surface_num_clouds.restype = sz
        L.och_surface_cloud_sizes.argtypes = [vp, _u64p]
        L.och_surface_cloud_sizes.restype = None
        L.och_surface_set_clouds.argtypes = [vp, sz, _u64p, _f64p]
        L.och_surface_set_clouds.restype = None
        L.och_checkpoint_validate.argtypes = [C.c_char_p]
        L.och_checkpoint_save.argtypes = [C.c_char_p, vp, C.POINTER(vp), sz, C.c_char_p, _f64p]
        L.och_checkpoint_load.argtypes = [C.c_char_p, vp]
        L.och_checkpoint_load.restype = vp
        L.och_checkpoint_destroy.argtypes = [vp]
        L.och_checkpoint_destroy.restype = None
        L.och_checkpoint_num_surfaces.argtypes = [vp]
        L.och_checkpoint_num_surfaces.restype = sz
        L.och_checkpoint_state.argtypes = [vp]
        L.och_checkpoint_state.restype = C.c_char_p
        L.och_checkpoint_info.argtypes = [vp, _f64p]
        L.och_checkpoint_info.restype = None
        L.och_checkpoint_get_surface.argtypes = [vp, sz, vp]
        L.och_densify_mesh.argtypes = [vp, vp, vp, _f64p, vp, sz]
        L.och_refine_by_point_density.argtypes = [vp, sz, C.c_double, C.c_int, C.c_double]
        L.och_refine_by_point_density.restype = sz
        L.och_refine_at_point.argtypes = [vp, C.c_double, C.c_double, C.c_int]
        L.och_refine_at_point.restype = sz
        L.och_count_points_per_triangle.argtypes = [vp, _u64p, _f64p, sz]
        L.och_count_points_per_triangle.restype = sz
        L.och_surface_locate.argtypes = [vp, _f64p, sz, _u64p]
        L.och_surface_locate.restype = None
        L.och_mesh_refinement_run.argtypes = [vp, vp, vp, C.c_int, _f64p]
        L.och_shard_block.argtypes = [u32, u32, u32, C.POINTER(u32), C.POINTER(u32)]
        L.och_shard_block.restype = None
        L.och_shard_begin.restype = vp
        L.och_shard_begin.argtypes = [vp, vp, u32, u32, _f64p, vp, u32, u32, _u64p]
        L.och_shard_destroy.argtypes = [vp]
        L.och_shard_destroy.restype = None
        L.och_shard_counts.argtypes = [vp, _u64p]
        L.och_shard_counts.restype = None
        L.och_shard_load_link_local.argtypes = [vp, vp, C.c_int, C.c_int, u32, C.c_int]
        for name in ("och_shard_subsets_export", "och_shard_edges_export"):
            getattr(L, name).argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
        for name in ("och_shard_subsets_import", "och_shard_edges_import"):
            getattr(L, name).argtypes = [vp, vp, u64]
        L.och_shard_link_remote.argtypes = [vp]
        L.och_shard_finalize.argtypes = [vp, _f64p, _f64p, _f64p]
        L.och_hilbert_xy2d.argtypes = [C.c_int, C.c_int, C.c_int]
        L.och_hilbert_xy2d.restype = u32
        _lib = L
    return _lib


def extract_features_batch(ctx, images_bgr, max_keypoints=20000, device_shape=None):
    """extract_features for a batch of equally sized BGR images on the device.  images_bgr: (n, h, w, 3) uint8
    host array, or - with device_shape=(n, h, w) - an integer device pointer to images already in HBM.  Returns
    a list of (loc [k x 2] f64, strength [k] f32, desc [k x 8] u64, num_sparse) per image."""
    L = load()
    if device_shape is None:
        imgs = np.ascontiguousarray(images_bgr, np.uint8)
        n, h, w, _ = imgs.shape
        src, on_dev = imgs.ctypes.data, 0
    else:
        n, h, w = device_shape
        src, on_dev = int(images_bgr), 1
    max_out = max_keypoints + 1   # the NMS seed re-enters the dense list (extract_features.cpp:63-83)
    loc = np.zeros((n, max_out, 2))
    st = np.zeros((n, max_out), np.float32)
    de = np.zeros((n, max_out, 8), np.uint64)
    counts, ns = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    rc = L.och_extract_features_batch(ctx.h, src, n, w, h, max_keypoints, max_out, loc.ctypes.data,
                                      st.ctypes.data, de.ctypes.data, counts.ctypes.data, ns.ctypes.data, on_dev)
    if rc != 0:
        raise capi.OchipError("extract failed: " + L.och_extract_last_error().decode())
    return [(loc[i, :counts[i]].copy(), st[i, :counts[i]].copy(), de[i, :counts[i]].copy(), int(ns[i])) for i in range(n)]


RELAX_SUMMARY_NAMES = ["solves", "iterations_total", "last_iterations", "initial_cost", "final_cost", "residual_blocks",
                       "setup_host_s", "device_s"]


def pack_edges(edges):
    """edges: list of dicts {src, dst, H (3x3) or None, px (k x 4), match_index (k,), dist (m,) or None}.
    Returns the flat arrays the stand-alone relax entry points (och_relax_ground_plane, och_relax) take."""
    n = len(edges)
    src = np.array([e["src"] for e in edges], np.uint64)
    dst = np.array([e["dst"] for e in edges], np.uint64)
    H = np.full((max(n, 1), 9), np.nan)
    ish = np.zeros(max(n, 1), np.uint8)
    for i, e in enumerate(edges):
        if e.get("H") is not None:
            H[i] = np.asarray(e["H"], np.float64).reshape(9)
            ish[i] = 1
    counts = [len(e["px"]) for e in edges]
    inl_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    px = np.ascontiguousarray(np.concatenate([np.asarray(e["px"], np.float64).reshape(-1, 4) for e in edges])
                              if n and sum(counts) else np.zeros((1, 4)))
    mi = np.ascontiguousarray(np.concatenate([np.asarray(e["match_index"], np.uint64) for e in edges])
                              if n and sum(counts) else np.zeros(1, np.uint64))
    dcounts = [0 if e.get("dist") is None else len(e["dist"]) for e in edges]
    dist_off = np.concatenate([[0], np.cumsum(dcounts)]).astype(np.uint64)
    dist = np.ascontiguousarray(np.concatenate([np.asarray(e["dist"], np.float64) for e in edges if e.get("dist") is not None])
                                if sum(dcounts) else np.zeros(1))
    return dict(src=src, dst=dst, H=np.ascontiguousarray(H), is_h=ish, inl_off=inl_off, px=px, match_index=mi,
                dist_off=dist_off, dist=dist)


RELAX_OPTIONS = dict(ORIENTATION=1 << 0, POSITION=1 << 1, GROUND_PLANE=1 << 2, GROUND_MESH=1 << 3, POINTS_3D=1 << 4,
                     FOCAL_LENGTH=1 << 5, PRINCIPAL_POINT=1 << 6, LENS_DISTORTIONS_RADIAL=1 << 7, BROWN2=1 << 8, BROWN24=1 << 9,
                     BROWN246=1 << 10, LENS_DISTORTIONS_TANGENTIAL=1 << 11, MINIMAL_MESH=1 << 12)
RELAX_SUMMARY12 = ["solves", "iterations_total", "last_iterations", "initial_cost", "final_cost", "residual_blocks",
                   "setup_host_s", "device_s", "track_blocks", "two_ray_blocks", "mesh_vertices", "unknowns"]


def relax_options(*names):
    bits = 0
    for n in names:
        bits |= RELAX_OPTIONS[n]
    return bits


class Surface:
    """surface_model of the host library: mesh (vertices, edges {source, dest, border, opposite 0, opposite 1}) + cloud."""

    def __init__(self):
        self.L = load()
        self.h = self.L.och_surface_create()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.och_surface_destroy(self.h)
            self.h = None

    def arrays(self):
        nv, ne, nc = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        self.L.och_surface_counts(self.h, C.byref(nv), C.byref(ne), C.byref(nc))
        v, e, c = np.zeros((max(nv.value, 1), 3)), np.zeros((max(ne.value, 1), 5), np.uint64), np.zeros((max(nc.value, 1), 3))
        self.L.och_surface_get(self.h, v.ctypes.data, e.ctypes.data, c.ctypes.data)
        return dict(vertices=v[:nv.value], edges=e[:ne.value], cloud=c[:nc.value])

    def set(self, vertices, edges, cloud=None):
        v = np.ascontiguousarray(vertices, np.float64).reshape(-1, 3)
        e = np.ascontiguousarray(edges, np.uint64).reshape(-1, 5)
        c = np.zeros((0, 3)) if cloud is None else np.ascontiguousarray(cloud, np.float64).reshape(-1, 3)
        pad = lambda a, shape, dt: a if len(a) else np.zeros(shape, dt)
        self.L.och_surface_set(self.h, len(v), pad(v, (1, 3), np.float64), len(e), pad(e, (1, 5), np.uint64), len(c),
                               pad(c, (1, 3), np.float64))
        return self


    def set_heights(self, z):
        """New vertex heights; the mesh's topology and container orders stay (what a relax does to a mesh)."""
        self.L.och_surface_set_heights(self.h, np.ascontiguousarray(z, np.float64))
        return self

    def clouds(self):
        """The surface's point clouds one by one (arrays() concatenates them)."""
        n = self.L.och_surface_num_clouds(self.h)
        sizes = np.zeros(max(n, 1), np.uint64)
        self.L.och_surface_cloud_sizes(self.h, sizes)
        pts, out, k = self.arrays()["cloud"], [], 0
        for i in range(n):
            out.append(pts[k:k + int(sizes[i])].copy())
            k += int(sizes[i])
        return out

    def set_clouds(self, clouds):
        sizes = np.array([len(c) for c in clouds] or [0], np.uint64)
        xyz = np.ascontiguousarray(np.concatenate([np.asarray(c, np.float64).reshape(-1, 3) for c in clouds])
                                   if len(clouds) and sizes.sum() else np.zeros((1, 3)))
        self.L.och_surface_set_clouds(self.h, len(clouds), sizes, xyz)
        return self

    def refine_by_point_density(self, max_points_per_triangle, min_distance_variance=0.0, max_iterations=10, min_triangle_size=0.0):
        """refineByPointDensity (src/surface/refine_mesh.cpp:827-909) with the surface's own clouds; returns triangles created."""
        return self.L.och_refine_by_point_density(self.h, max_points_per_triangle, min_distance_variance, max_iterations,
                                                  min_triangle_size)

    def refine_at_point(self, x, y, levels=1):
        return self.L.och_refine_at_point(self.h, x, y, levels)

    def count_points_per_triangle(self):
        """countPointsPerTriangle: (vertices n x 3, counts n, distance variances n) in first-point order."""
        cap = 2 * max(len(self.arrays()["edges"]), 1)
        tri, st = np.zeros((cap, 3), np.uint64), np.zeros((cap, 2))
        n = self.L.och_count_points_per_triangle(self.h, tri, st, cap)
        return tri[:n], st[:n, 0].astype(np.int64), st[:n, 1]

    def locate(self, xy):
        xy = np.ascontiguousarray(xy, np.float64).reshape(-1, 2)
        tri = np.zeros((max(len(xy), 1), 3), np.uint64)
        self.L.och_surface_locate(self.h, xy if len(xy) else np.zeros((1, 2)), len(xy), tri)
        return tri[:len(xy)]

    def save_ply(self, path):
        """serialize(MeshGraph, ostream) of the reference (ASCII PLY, src/io/serialize_MeshGraph.cpp)."""
        if self.L.och_surface_save_ply(self.h, str(path).encode()) != 0:
            raise IOError("cannot write %s" % path)

    def load_ply(self, path):
        if self.L.och_surface_load_ply(self.h, str(path).encode()) != 0:
            raise IOError("%s is not a surface PLY of the reference's layout" % path)
        return self


def validate_checkpoint(path):
    """validateCheckpoint (src/io/checkpoint.cpp:318-337)."""
    return bool(load().och_checkpoint_validate(str(path).encode()))


def save_checkpoint(path, graph, surfaces=(), state="INITIAL_PROCESSING", state_run_count=0, origin=(0.0, 0.0)):
    """saveCheckpoint (src/io/checkpoint.cpp:155-231): metadata.json, graph.json, surface_<i>.ply, pointcloud_<i>_<j>.xyz."""
    L = load()
    arr = (C.c_void_p * max(len(surfaces), 1))(*[s.h for s in surfaces])
    info = np.array([state_run_count, origin[0], origin[1], 0.0])
    if L.och_checkpoint_save(str(path).encode(), graph.h, arr, len(surfaces), state.encode(), info) != 0:
        raise IOError(L.och_last_error(graph.h).decode())


def load_checkpoint(path):
    """loadCheckpoint (src/io/checkpoint.cpp:233-316).  Returns (Graph, [Surface], info dict)."""
    L = load()
    g = Graph()
    cp = L.och_checkpoint_load(str(path).encode(), g.h)
    if not cp:
        raise IOError(L.och_last_error(g.h).decode())
    g._refresh_ids()
    try:
        surfaces = []
        for i in range(L.och_checkpoint_num_surfaces(cp)):
            s = Surface()
            L.och_checkpoint_get_surface(cp, i, s.h)
            surfaces.append(s)
        info = np.zeros(4)
        L.och_checkpoint_info(cp, info)
        meta = dict(state=L.och_checkpoint_state(cp).decode(), state_run_count=int(info[0]), origin=(info[1], info[2]))
    finally:
        L.och_checkpoint_destroy(cp)
    return g, surfaces, meta


def rebuild_mesh(cam_xyz, previous=None, minimal=False):
    """rebuildMesh / buildMinimalMesh of the host library (no device involved)."""
    cam_xyz = np.ascontiguousarray(cam_xyz, np.float64).reshape(-1, 3)
    s = Surface()
    s.L.och_rebuild_mesh(cam_xyz, len(cam_xyz), previous.h if previous is not None else None, int(minimal), s.h)
    return s


def relax(ctx, node_pos, node_ori, model10, features, pose_node, pose_ori, packed_edges, options, grid_fraction=0.1,
          opt_edges=None, previous=None, cam_model=None, edge_poses=None, points_mode=-1):
    """relax(graph, nodes, cam_models, edges, config, previousSurfaces) on the device, any flavour.  features: per node
    an (k x 2) array of feature locations; packed_edges as for relax_ground_plane plus 'feat' (inliers x 2 feature
    indices).  edge_poses (n_edges x 4 x 8): the edges' homography decompositions (relative-orientation flavour);
    points_mode 0 / 1 / 2 with POINTS_3D: set up the 3-D point problem and leave it / solve it / run
    relaxObservedModelOnly, returning the points before and after (test/test_relax.cpp:470-483)."""
    L = load()
    node_pos = np.ascontiguousarray(node_pos, np.float64)
    node_ori = np.ascontiguousarray(node_ori, np.float64)
    pose_node = np.ascontiguousarray(pose_node, np.uint64)
    pose_ori = np.ascontiguousarray(pose_ori, np.float64).copy()
    feat_off = np.concatenate([[0], np.cumsum([len(f) for f in features])]).astype(np.uint64)
    feat_xy = np.ascontiguousarray(np.concatenate([np.asarray(f, np.float64).reshape(-1, 2) for f in features])
                                   if feat_off[-1] else np.zeros((1, 2)))
    pk = packed_edges
    n_edges = len(pk["src"])
    opt = np.ascontiguousarray(np.arange(n_edges) if opt_edges is None else opt_edges, np.uint64)
    summary = np.zeros(12)
    out_surface = Surface()
    cm = np.ascontiguousarray(model10 if cam_model is None else cam_model, np.float64).copy()
    ep = None if edge_poses is None else np.ascontiguousarray(edge_poses, np.float64).reshape(-1, 32)
    cap = 1 << 20
    before = after = None
    n_pts = C.c_size_t(0)
    if points_mode >= 0:
        before, after = np.zeros((cap, 3)), np.zeros((cap, 3))
    rc = L.och_relax_ex(ctx.h, len(node_pos), node_pos, node_ori, np.ascontiguousarray(model10, np.float64), feat_off, feat_xy,
                        len(pose_node), pose_node, pose_ori, n_edges, pk["src"], pk["dst"], pk["H"].ctypes.data, pk["is_h"],
                        pk["inl_off"], pk["px"], np.ascontiguousarray(pk["feat"], np.uint64), pk["match_index"],
                        pk["dist_off"].ctypes.data, pk["dist"].ctypes.data, len(opt), opt if len(opt) else np.zeros(1, np.uint64),
                        options, grid_fraction, previous.h if previous is not None else None, out_surface.h, summary,
                        cm.ctypes.data, None if ep is None else ep.ctypes.data, points_mode,
                        None if before is None else before.ctypes.data, None if after is None else after.ctypes.data,
                        cap if before is not None else 0, C.byref(n_pts))
    if rc != 0:
        raise capi.OchipError("relax failed: " + L.och_relax_last_error().decode())
    out = dict(zip(RELAX_SUMMARY12, summary.tolist()))
    out.update(orientation=pose_ori, surface=out_surface, cam_model=cm)
    if before is not None:
        out.update(points_before=before[:n_pts.value].copy(), points_after=after[:n_pts.value].copy())
    for k in ("solves", "iterations_total", "last_iterations", "residual_blocks", "track_blocks", "two_ray_blocks",
              "mesh_vertices", "unknowns"):
        out[k] = int(out[k])
    return out


def relax_setup_check(on=-1):
    """Test hook (och_debug_relax_setup_check): on=1 makes every ground-plane relax set-up also run the host's grid filter
    and block assembly and fail unless the device's blocks equal them bit for bit.  Returns the set-ups compared so far."""
    return int(load().och_debug_relax_setup_check(int(on)))


def relax_ground_plane(ctx, node_pos, node_ori, model10, pose_node, pose_ori, packed_edges, opt_edges=None):
    """relax(graph, nodes, cam_models, edges, {ORIENTATION, GROUND_PLANE}, {}) on the device.  `packed_edges` is
    the dict of flat arrays (src, dst, H, is_h, inl_off, px, match_index, dist_off, dist)."""
    L = load()
    node_pos = np.ascontiguousarray(node_pos, np.float64)
    node_ori = np.ascontiguousarray(node_ori, np.float64)
    pose_node = np.ascontiguousarray(pose_node, np.uint64)
    pose_ori = np.ascontiguousarray(pose_ori, np.float64).copy()
    pk = packed_edges
    n_edges = len(pk["src"])
    opt = np.ascontiguousarray(np.arange(n_edges) if opt_edges is None else opt_edges, np.uint64)
    plane, summary = np.zeros(9), np.zeros(8)
    rc = L.och_relax_ground_plane(ctx.h, len(node_pos), node_pos, node_ori, np.ascontiguousarray(model10, np.float64),
                                  len(pose_node), pose_node, pose_ori, n_edges, pk["src"], pk["dst"], pk["H"], pk["is_h"],
                                  pk["inl_off"], pk["px"], pk["match_index"], pk["dist_off"].ctypes.data,
                                  pk["dist"].ctypes.data, len(opt), opt, plane, summary)
    if rc != 0:
        raise capi.OchipError("relax failed: " + L.och_relax_last_error().decode())
    out = dict(zip(RELAX_SUMMARY_NAMES, summary.tolist()))
    out.update(orientation=pose_ori, plane=plane.reshape(3, 3))
    for k in ("solves", "iterations_total", "last_iterations", "residual_blocks"):
        out[k] = int(out[k])
    return out


LINK_TIMER_NAMES = ["link_init", "subsample", "upload", "match_device", "match_host", "ransac_device",
                    "decompose_host", "link_finalize"]


class Graph:
    """MeasurementGraph + stage drivers (opencalibration_amd/csrc/host)."""

    def __init__(self):
        self.L = load()
        self.h = C.c_void_p(self.L.och_graph_create())
        self.node_ids = []

    def close(self):
        if getattr(self, "h", None):
            self.L.och_graph_destroy(self.h)
            self.h = None

    # ---- graph.json (src/io/serialize_MeasurementGraph.cpp, src/io/deserialize_MeasurementGraph.cpp)
    def to_json(self):
        n = C.c_size_t(0)
        ptr = self.L.och_graph_to_json(self.h, C.byref(n))
        if not ptr:
            raise MemoryError("serialize failed")
        try:
            return C.string_at(ptr, n.value).decode("utf-8")
        finally:
            self.L.och_free(ptr)

    def _refresh_ids(self):
        ids = np.zeros(max(self.L.och_graph_num_nodes(self.h), 1), np.uint64)
        self.L.och_graph_node_ids(self.h, ids)
        self.node_ids = [int(i) for i in ids[:self.L.och_graph_num_nodes(self.h)]]

    def from_json(self, text):
        """deserialize(json, graph): replaces this graph.  Raises ValueError on anything but a version-1 document."""
        raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        if self.L.och_graph_from_json(self.h, raw, len(raw)) != 0:
            raise ValueError(self.L.och_last_error(self.h).decode())
        self._refresh_ids()
        return self

    def save_json(self, path):
        if self.L.och_graph_save_json(self.h, str(path).encode()) != 0:
            raise IOError(self.L.och_last_error(self.h).decode())

    def load_json(self, path):
        if self.L.och_graph_load_json(self.h, str(path).encode()) != 0:
            raise ValueError(self.L.och_last_error(self.h).decode())
        self._refresh_ids()
        return self

    def densify_mesh(self, ctx, surface, want_matches=False, match_cap=1 << 22):
        """densifyMesh (src/dense/dense_stereo.cpp:66-403): dense guided matching on the device against `surface`'s mesh;
        the triangulated points become one more cloud of `surface`.  Returns the stats (and the accepted matches)."""
        stats = np.zeros(10)
        pairs = np.zeros((match_cap if want_matches else 1, 2), np.uint64)
        rc = self.L.och_densify_mesh(self.h, ctx.h, surface.h, stats, pairs.ctypes.data if want_matches else None,
                                     match_cap if want_matches else 0)
        if rc != 0:
            raise capi.OchipError("densify failed: " + self.L.och_last_error(self.h).decode())
        names = ["images", "dense_features", "queries", "matches", "tracks", "points", "index_s", "rays_s", "device_s", "tracks_s"]
        out = dict(zip(names, stats.tolist()))
        for k in names[:6]:
            out[k] = int(out[k])
        if want_matches:
            out["match_pairs"] = pairs[:min(out["matches"], match_cap)].copy()
        return out

    def mesh_refinement(self, ctx, surface=None, max_steps=40):
        """The pipeline's MESH_REFINEMENT state (src/pipeline/pipeline.cpp:666-819) run to its end: returns (surface, log)."""
        surface = Surface() if surface is None else surface
        log = np.zeros((max_steps, 8))
        n = self.L.och_mesh_refinement_run(self.h, ctx.h, surface.h, max_steps, log)
        if n < 0:
            raise capi.OchipError("mesh refinement failed: " + self.L.och_last_error(self.h).decode())
        names = ["level", "grid_fraction", "gsd", "above_threshold", "max_points", "created", "vertices", "repeat"]
        return surface, [dict(zip(names, row)) for row in log[:n].tolist()]

    def node_table(self):
        """Per node in graph order: id, index into models(), number of features, number of sparse features."""
        n = self.L.och_graph_num_nodes(self.h)
        ids, mi = np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint32)
        nf, ns = np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint64)
        self.L.och_graph_node_table(self.h, ids.ctypes.data, mi.ctypes.data, nf.ctypes.data, ns.ctypes.data)
        return dict(id=ids[:n], model=mi[:n], features=nf[:n], sparse=ns[:n])

    def node_payload(self, index):
        """Features (loc, strength, desc), position, orientation and path of the index-th node."""
        n = int(self.node_table()["features"][index])
        loc, st, de = np.zeros((max(n, 1), 2)), np.zeros(max(n, 1), np.float32), np.zeros((max(n, 1), 8), np.uint64)
        pos, ori = np.zeros(3), np.zeros(4)
        self.L.och_graph_node_payload(self.h, index, loc.ctypes.data, st.ctypes.data, de.ctypes.data, pos.ctypes.data,
                                      ori.ctypes.data)
        return dict(loc=loc[:n], strength=st[:n], desc=de[:n], position=pos, orientation=ori,
                    path=self.L.och_graph_node_path(self.h, index).decode("utf-8"))

    def set_node_path(self, index, path):
        self.L.och_graph_set_node_path(self.h, index, path.encode("utf-8"))

    def models(self):
        """The graph's camera models: rows of och_graph_add_model's ten numbers followed by the model id."""
        out = np.zeros((self.L.och_graph_num_models(self.h), 11))
        for i in range(len(out)):
            self.L.och_graph_get_model(self.h, i, out[i])
        return out

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_model(self, model10):
        return self.L.och_graph_add_model(self.h, np.ascontiguousarray(model10, np.float64))

    def set_model(self, model, model10):
        """Replace the intrinsics of camera model `model` (what a relax with free intrinsics writes back)."""
        if self.L.och_graph_set_model(self.h, int(model), np.ascontiguousarray(model10, np.float64)) != 0:
            raise capi.OchipError("och_graph_set_model: bad model index")

    def refit_edges(self, ctx):
        """RelaxGroup::finalize's edge loop after a model change (relax_group.cpp:137-177): every edge re-fitted on its
        previous inliers with the current camera models."""
        if self.L.och_graph_refit_edges(self.h, ctx.h) != 0:
            raise capi.OchipError("refit failed: " + self.L.och_last_error(self.h).decode())

    def add_image(self, loc, strength, desc, num_sparse, model, position):
        nid = self.L.och_graph_add_image(self.h, np.ascontiguousarray(loc, np.float64),
                                         np.ascontiguousarray(strength, np.float32),
                                         np.ascontiguousarray(desc, np.uint64), len(strength), int(num_sparse), model,
                                         np.ascontiguousarray(position, np.float64))
        self.node_ids.append(nid)
        return nid

    def load_images(self, ctx, images_bgr, model, positions, max_keypoints=30000, device_shape=None):
        """The load stage for a batch of equally sized images: extract_features on the device, one node per image.
        images_bgr: (n, h, w, 3) uint8 host array or, with device_shape=(n, h, w), a device pointer.  Returns
        (mean features per image, mean sparse features per image)."""
        if device_shape is None:
            imgs = np.ascontiguousarray(images_bgr, np.uint8)
            n, h, w, _ = imgs.shape
            src, on_dev = imgs.ctypes.data, 0
        else:
            n, h, w = device_shape
            src, on_dev = int(images_bgr), 1
        ids, totals = np.zeros(max(n, 1), np.uint64), np.zeros(2)
        rc = self.L.och_graph_load_images(self.h, ctx.h, src, n, w, h, max_keypoints, on_dev, model,
                                          np.ascontiguousarray(positions, np.float64).reshape(-1, 3), ids, totals)
        if rc != 0:
            raise capi.OchipError("load_images failed: " + self.L.och_last_error(self.h).decode())
        self.node_ids += [int(i) for i in ids[:n]]
        return totals[0] / max(n, 1), totals[1] / max(n, 1)

    def load_link_images(self, ctx, images_bgr, model, positions, orientations=None, max_keypoints=30000, device_shape=None):
        """Load and link overlapped (och_graph_load_link_images): extraction streams in chunks and ranges of links run
        on their own device contexts as soon as their images are ready.  Same graph as load_images() + link().
        Returns (mean features per image, mean sparse features per image, link timers, (extract_s, total_s))."""
        if device_shape is None:
            imgs = np.ascontiguousarray(images_bgr, np.uint8)
            n, h, w, _ = imgs.shape
            src, on_dev = imgs.ctypes.data, 0
        else:
            n, h, w = device_shape
            src, on_dev = int(images_bgr), 1
        ids, totals, timers, stage = np.zeros(max(n, 1), np.uint64), np.zeros(2), np.zeros(8), np.zeros(2)
        ori = None if orientations is None else np.ascontiguousarray(orientations, np.float64)
        rc = self.L.och_graph_load_link_images(self.h, ctx.h, src, n, w, h, max_keypoints, on_dev, model,
                                               np.ascontiguousarray(positions, np.float64).reshape(-1, 3),
                                               None if ori is None else ori.ctypes.data, ids, totals, timers, stage)
        if rc != 0:
            raise capi.OchipError("load_link_images failed: " + self.L.och_last_error(self.h).decode())
        self.node_ids += [int(i) for i in ids[:n]]
        return totals[0] / max(n, 1), totals[1] / max(n, 1), dict(zip(LINK_TIMER_NAMES, timers.tolist())), (stage[0], stage[1])

    def initial_processing(self, ctx):
        """Pipeline::Impl::initial_processing's stepper (och_initial_processing_*): see InitialProcessing."""
        return InitialProcessing(self, ctx)

    @classmethod
    def from_synthetic(cls, grid):
        g = cls()
        m = g.add_model(grid.model)
        for i in range(grid.n_images):
            loc, st, de, _ = grid.image(i)
            g.add_image(loc, st, de, grid.num_sparse[i], m, grid.position[i])
        return g

    def add_edge(self, source_id, dest_id, px, f1, f2, match_index=None, H=None, dist=None, poses=None, match_idx=None,
                 is_homography=None):
        """graph.addEdge from arrays: px n x 4 inlier pixels, f1 / f2 feature indices, match distances (and the matches'
        feature index pairs)."""
        px = np.ascontiguousarray(px, np.float64).reshape(-1, 4)
        n = len(px)
        idx = np.zeros((max(n, 1), 3), np.uint64)
        idx[:n, 0], idx[:n, 1] = f1, f2
        idx[:n, 2] = np.arange(n) if match_index is None else match_index
        Hc = None if H is None else np.ascontiguousarray(H, np.float64)
        d = None if dist is None or len(dist) == 0 else np.ascontiguousarray(dist, np.float64)
        pc = None if poses is None else np.ascontiguousarray(poses, np.float64)
        mi2 = None if match_idx is None or d is None else np.ascontiguousarray(match_idx, np.uint64).reshape(-1, 2)
        e = self.L.och_graph_add_edge(self.h, int(source_id), int(dest_id), None if Hc is None else Hc.ctypes.data,
                                      int(H is not None if is_homography is None else is_homography), n,
                                      px if n else np.zeros((1, 4)), idx, 0 if d is None else len(d),
                                      None if mi2 is None else mi2.ctypes.data, None if d is None else d.ctypes.data,
                                      None if pc is None else pc.ctypes.data)
        if e == 0:
            raise capi.OchipError(self.L.och_last_error(self.h).decode())
        return e

    def match_work(self):
        """{pairs, distances needed, subset features} of the last link stage's match step."""
        out = np.zeros(3)
        self.L.och_link_match_work(self.h, out)
        return dict(pairs=out[0], distances=out[1], subset_features=out[2])

    def orientations(self):
        out = np.zeros((max(self.num_nodes, 1), 4))
        self.L.och_graph_get_orientations(self.h, out)
        return out[:self.num_nodes]

    @property
    def num_nodes(self):
        return self.L.och_graph_num_nodes(self.h)

    @property
    def num_edges(self):
        return self.L.och_graph_num_edges(self.h)

    def link(self, ctx, node_ids=None, keep_debug=False):
        """LinkStage init -> runner -> finalize on the device owned by `ctx`; returns the stage timers."""
        ids = np.ascontiguousarray(self.node_ids if node_ids is None else node_ids, np.uint64)
        timers = np.zeros(8)
        rc = self.L.och_link_stage_run(self.h, ctx.h, ids, len(ids), int(keep_debug), timers)
        if rc != 0:
            raise capi.OchipError("link stage failed: " + self.L.och_last_error(self.h).decode())
        return dict(zip(LINK_TIMER_NAMES, timers.tolist()))

    def relax_ground_plane(self, ctx, orientations, shard=None):
        """All nodes as one relax group, every edge whitelisted; updates and returns the orientations.
        shard = (rank, world, exchange): evaluate only this rank's share of the residual blocks; `exchange` is a
        RELAX_EXCHANGE_FN (parallel.relax_exchange builds one on torch.distributed) or a capi.RcclComm created on `ctx`
        (the native transport: RCCL all-gathers on the context's stream), and every rank gets the same, bit-identical
        result."""
        ori = np.ascontiguousarray(orientations, np.float64).copy()
        if ori.shape != (self.num_nodes, 4):
            raise ValueError("orientations must be %d x 4, got %r" % (self.num_nodes, ori.shape))
        plane, summary = np.zeros(9), np.zeros(8)
        if shard is None:
            rc = self.L.och_graph_relax_ground_plane(self.h, ctx.h, ori, plane, summary)
        else:
            rank, world, exchange = shard
            if isinstance(exchange, capi.RcclComm):
                fn, user = exchange.exchange
            else:
                fn, user = C.cast(exchange, C.c_void_p).value, None
            rc = self.L.och_graph_relax_ground_plane_sharded(self.h, ctx.h, ori, plane, summary, rank, world, fn, user)
        if rc != 0:
            raise capi.OchipError("relax failed: " + self.L.och_last_error(self.h).decode())
        out = dict(zip(RELAX_SUMMARY_NAMES, summary.tolist()))
        out.update(orientation=ori, plane=plane.reshape(3, 3))
        return out

    def relax(self, ctx, orientations, options, grid_fraction=0.1, previous=None, shard=None):
        """All nodes as one relax group, every edge whitelisted, any flavour (options: relax_options(...)).
        shard = (rank, world, exchange) as for relax_ground_plane: the residual blocks' evaluation over the ranks."""
        ori = np.ascontiguousarray(orientations, np.float64).copy()
        summary = np.zeros(12)
        surface = Surface()
        prev = previous.h if previous is not None else None
        if shard is None:
            rc = self.L.och_graph_relax(self.h, ctx.h, ori, options, grid_fraction, prev, surface.h, summary)
        else:
            rank, world, exchange = shard
            if isinstance(exchange, capi.RcclComm):
                fn, user = exchange.exchange
            else:
                fn, user = C.cast(exchange, C.c_void_p).value, None
            rc = self.L.och_graph_relax_sharded(self.h, ctx.h, ori, options, grid_fraction, prev, surface.h, summary, rank, world,
                                                fn, user)
        if rc != 0:
            raise capi.OchipError("relax failed: " + self.L.och_last_error(self.h).decode())
        out = dict(zip(RELAX_SUMMARY12, summary.tolist()))
        out.update(orientation=ori, surface=surface)
        return out

    def relax_stage(self, ctx, options, grid_fraction=0.1, node_ids=None, disable_parallelism=False, max_groups=0,
                    previous=None, shard=None):
        """RelaxStage::init + runners + finalize (relax_stage.cpp): node_ids None = relax_all.  Returns the summary, the
        merged surface, the graph's orientations afterwards and the group of every node (-1: not a primary node).
        shard = (rank, world, all_gather): this rank runs groups rank, rank + world, ... and all_gather(bytes) -> list of
        every rank's bytes (parallel.all_gather_bytes) carries the groups' results; every rank ends with the
        single-process result."""
        ids = None if node_ids is None else np.ascontiguousarray(node_ids, np.uint64)
        summary = np.zeros(13)
        groups = np.full(max(self.num_nodes, 1), -1, np.int64)
        surface = Surface()
        if shard is None:
            rc = self.L.och_relax_stage_run(self.h, ctx.h, None if ids is None else ids.ctypes.data, 0 if ids is None else len(ids),
                                            int(ids is None), int(disable_parallelism), options, grid_fraction, max_groups,
                                            previous.h if previous is not None else None, surface.h, groups, summary)
        else:
            rank, world, all_gather = shard
            st = self.L.och_relax_stage_begin(self.h, None if ids is None else ids.ctypes.data, 0 if ids is None else len(ids),
                                              int(ids is None), int(disable_parallelism), options, grid_fraction, max_groups,
                                              previous.h if previous is not None else None, groups)
            if not st:
                raise MemoryError("och_relax_stage_begin")
            self.L.och_relax_stage_run_groups(st, ctx.h, rank, world)
            ptr, n = C.c_void_p(0), C.c_uint64(0)
            self.L.och_relax_stage_export(st, rank, world, C.byref(ptr), C.byref(n))
            mine = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n.value,)) if n.value else np.zeros(0, np.uint8)
            rc = 0
            for r, part in enumerate(all_gather(mine)):
                if r != rank and len(part):
                    part = np.ascontiguousarray(part, np.uint8)
                    rc = rc or self.L.och_relax_stage_import(st, part.ctypes.data, len(part))
            rc = self.L.och_relax_stage_end(st, surface.h, summary) or rc
        if rc != 0:
            raise capi.OchipError("relax stage failed: " + self.L.och_last_error(self.h).decode())
        out = dict(zip(RELAX_SUMMARY12 + ["groups"], summary.tolist()))
        out.update(surface=surface, group_of_node=groups[:self.num_nodes].copy())
        return out

    def relax_partition(self, num_groups, ordered=False):
        """The spectral / k-means partition of RelaxStage::init alone (host only)."""
        groups, pos = np.full(max(self.num_nodes, 1), -1, np.int64), np.full(max(self.num_nodes, 1), -1, np.int64)
        n = self.L.och_relax_partition(self.h, num_groups, groups, pos)
        groups, pos = groups[:self.num_nodes].copy(), pos[:self.num_nodes].copy()
        if ordered:
            return [[int(i) for i in sorted(np.flatnonzero(groups == g), key=lambda i: pos[i])] for g in range(n)]
        return n, groups

    def link_debug(self):
        out = []
        for p in range(self.L.och_link_debug_count(self.h)):
            ids, n, score, it = np.zeros(2, np.uint64), np.zeros(1, np.uint64), np.zeros(1), np.zeros(3, np.uint32)
            self.L.och_link_debug_pair(self.h, p, ids, n, score, it)
            m = max(int(n[0]), 1)
            i1, i2, d, inl = np.zeros(m, np.uint64), np.zeros(m, np.uint64), np.zeros(m), np.zeros(m, np.uint8)
            self.L.och_link_debug_matches(self.h, p, i1, i2, d, inl)
            k = int(n[0])
            out.append(dict(node=int(ids[0]), match_node=int(ids[1]), i1=i1[:k], i2=i2[:k], dist=d[:k], inliers=inl[:k],
                            score=float(score[0]), iterations=int(it[0]), improvements=int(it[1]),
                            can_decompose=bool(it[2])))
        return out

    def edges(self, with_distances=False):
        out = []
        for e in range(self.num_edges):
            ids, cnt, H, poses = np.zeros(2, np.uint64), np.zeros(2, np.uint64), np.zeros((3, 3)), np.zeros((4, 8))
            self.L.och_graph_edge_info(self.h, e, ids, cnt, H, poses)
            k = max(int(cnt[1]), 1)
            f1, f2, mi, px = np.zeros(k, np.uint64), np.zeros(k, np.uint64), np.zeros(k, np.uint64), np.zeros((k, 4))
            self.L.och_graph_edge_inliers(self.h, e, f1, f2, mi, px)
            k = int(cnt[1])
            dist = np.zeros(max(int(cnt[0]), 1))
            midx = np.zeros((max(int(cnt[0]), 1), 2), np.uint64)
            is_h = C.c_int(0)
            if with_distances:
                self.L.och_graph_edge_matches(self.h, e, midx.ctypes.data, dist.ctypes.data, C.byref(is_h))
            out.append(dict(source=int(ids[0]), dest=int(ids[1]), n_matches=int(cnt[0]), n_inliers=k, H=H, poses=poses,
                            f1=f1[:k], f2=f2[:k], match_index=mi[:k], px=px[:k], dist=dist[:int(cnt[0])],
                            match_idx=midx[:int(cnt[0])], is_homography=bool(is_h.value)))
        return out

    def set_orientations(self, ori):
        self.L.och_graph_set_orientations(self.h, np.ascontiguousarray(ori, np.float64))

    def edges_flat(self, node_subset=None):
        """The linked edges as flat dicts (src/dst node index, H, inlier pixel pairs, match indices, match
        distances): the form the stand-alone relax entry points and the parity checker take; optionally
        restricted to edges inside a node-index subset."""
        index_of = {nid: i for i, nid in enumerate(self.node_ids)}
        keep = None if node_subset is None else set(int(i) for i in node_subset)
        out = []
        for ed in self.edges(with_distances=True):
            s, d = index_of[ed["source"]], index_of[ed["dest"]]
            if keep is not None and (s not in keep or d not in keep):
                continue
            out.append(dict(src=s, dst=d, H=ed["H"], px=ed["px"], match_index=ed["match_index"], dist=ed["dist"]))
        return out


SHARD_SECONDS = ["extract", "block_linked", "subsets_export", "subsets_import", "remote_links", "edges_export",
                 "edges_import", "finalize", "stage"]


def shard_block(n_images, rank, world):
    """(first, count) of the contiguous image block of `rank` (och_shard_block)."""
    a, b = C.c_uint32(0), C.c_uint32(0)
    load().och_shard_block(n_images, rank, world, C.byref(a), C.byref(b))
    return a.value, b.value


class Shard:
    """One survey's load + link stages on rank `rank` of `world` (include/oc_host.h, och_shard_*): the caller moves the two
    buffers between the ranks (parallel.survey_sharded does it with torch.distributed)."""

    def __init__(self, graph, ctx, model, positions, orientations, rank, world):
        self.g, self.L = graph, graph.L
        pos = np.ascontiguousarray(positions, np.float64).reshape(-1, 3)
        ori = None if orientations is None else np.ascontiguousarray(orientations, np.float64)
        ids = np.zeros(max(len(pos), 1), np.uint64)
        self.h = self.L.och_shard_begin(graph.h, ctx.h, len(pos), model, pos, None if ori is None else ori.ctypes.data, rank, world, ids)
        if not self.h:
            raise capi.OchipError("och_shard_begin: " + self.L.och_last_error(graph.h).decode())
        graph.node_ids += [int(i) for i in ids[:len(pos)]]
        self.first, self.count = shard_block(len(pos), rank, world)

    def close(self):
        if getattr(self, "h", None):
            self.L.och_shard_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc, what):
        if rc != 0:
            raise capi.OchipError(what + " failed: " + self.L.och_last_error(self.g.h).decode())

    def counts(self):
        out = np.zeros(4, np.uint64)
        self.L.och_shard_counts(self.h, out)
        return dict(zip(["images", "pairs_in_block", "pairs_across_blocks", "halo_images"], (int(v) for v in out)))

    def load_link_local(self, images_block, width, height, max_keypoints=30000, on_device=True):
        """images_block: device pointer (on_device) or (count, h, w, 3) uint8 host array of the BLOCK's images."""
        src = int(images_block) if on_device else np.ascontiguousarray(images_block, np.uint8).ctypes.data
        self._check(self.L.och_shard_load_link_local(self.h, src, width, height, max_keypoints, int(on_device)), "load + link of the block")

    def _export(self, fn):
        ptr, n = C.c_void_p(0), C.c_uint64(0)
        self._check(fn(self.h, C.byref(ptr), C.byref(n)), "export")
        if n.value == 0:
            return np.zeros(0, np.uint8)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n.value,))   # a view: copy before the next export

    def subsets_export(self):
        return self._export(self.L.och_shard_subsets_export)

    def edges_export(self):
        return self._export(self.L.och_shard_edges_export)

    @staticmethod
    def _aligned(buf):
        buf = np.ascontiguousarray(buf, np.uint8)
        if buf.ctypes.data % 8:
            tmp = np.zeros(len(buf) + 8, np.uint8)
            off = (-tmp.ctypes.data) % 8
            tmp[off:off + len(buf)] = buf
            buf = tmp[off:off + len(buf)]
        return buf

    def subsets_import(self, buf):
        buf = self._aligned(buf)
        self._check(self.L.och_shard_subsets_import(self.h, buf.ctypes.data, len(buf)), "subset import")

    def edges_import(self, buf):
        buf = self._aligned(buf)
        self._check(self.L.och_shard_edges_import(self.h, buf.ctypes.data, len(buf)), "edge import")

    def link_remote(self):
        self._check(self.L.och_shard_link_remote(self.h), "links across blocks")

    def finalize(self):
        """Returns (features, sparse features of the block, link timers, stage seconds)."""
        totals, timers, secs = np.zeros(2), np.zeros(8), np.zeros(9)
        self._check(self.L.och_shard_finalize(self.h, totals, timers, secs), "finalize")
        return totals[0], totals[1], dict(zip(LINK_TIMER_NAMES, timers.tolist())), dict(zip(SHARD_SECONDS, secs.tolist()))


def ransac_epipolar(ctx, model, rays, quality=None, threshold=0.01):
    """ransac<fundamental_matrix_model> (model 0) / ransac<essential_matrix_model> (model 1) on the device.  rays: n x 6
    {measurement1, measurement2}.  Returns (score, matrix 3 x 3, inliers, iterations, improvements)."""
    rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
    n = len(rays)
    M, inl, counts = np.zeros((3, 3)), np.zeros(max(n, 1), np.uint8), np.zeros(3, np.uint32)
    q = None if quality is None else np.ascontiguousarray(quality, np.float64)
    score = load().och_ransac_epipolar(ctx.h, model, rays if n else np.zeros((1, 6)), None if q is None else q.ctypes.data, n,
                                       float(threshold), M.reshape(9), inl, counts)
    if score != score and n:
        raise capi.OchipError("epipolar RANSAC failed: " + ctx.last_error() if hasattr(ctx, "last_error") else "epipolar RANSAC failed")
    return score, M, inl[:n].astype(bool), int(counts[0]), int(counts[1])


def extract_tail(kp6, desc, scale):
    """The host tail of extract_features on given keypoints (detection order): (loc, strength, desc, num_sparse)."""
    L = load()
    kp6 = np.ascontiguousarray(kp6, np.float32).reshape(-1, 6)
    desc = np.ascontiguousarray(desc, np.uint64).reshape(-1, 8)
    n = len(kp6)
    loc, st, d, ns = np.zeros((n + 1, 2)), np.zeros(n + 1, np.float32), np.zeros((n + 1, 8), np.uint64), np.zeros(1, np.uint64)  # the seed keypoint appears twice
    m = L.och_extract_tail(kp6 if n else np.zeros((1, 6), np.float32), desc if n else np.zeros((1, 8), np.uint64), n, float(scale), loc, st, d, ns)
    return loc[:m].copy(), st[:m].copy(), d[:m].copy(), int(ns[0])


def extract_tail_prepared(lists, scale, force_host_nms=False):
    """The host tail on lists the device prepared (capi.Context.feature_lists): (loc, strength, desc, num_sparse)."""
    L = load()
    n = len(lists["response"])
    loc, st, d, ns = np.zeros((n + 1, 2)), np.zeros(n + 1, np.float32), np.zeros((n + 1, 8), np.uint64), np.zeros(1, np.uint64)
    rec = np.ascontiguousarray(lists["records"], np.uint8) if n else np.zeros((1, 88), np.uint8)
    resp = np.ascontiguousarray(lists["response"], np.float32) if n else np.zeros(1, np.float32)
    slot = np.ascontiguousarray(lists["slot"], np.uint32) if n else np.zeros(1, np.uint32)
    m = L.och_extract_tail_prepared(rec.ctypes.data, resp, slot.ctypes.data, int(lists["num_sparse"]),
                                    int(bool(lists["conflict"]) or force_host_nms), n, float(scale), loc, st, d, ns)
    return loc[:m].copy(), st[:m].copy(), d[:m].copy(), int(ns[0])


def decompose(H, m1, m2):
    """homography_model::decompose on inlier rays m1, m2 (n x 3 each): (can_decompose, poses 4 x 8)."""
    rays = np.ascontiguousarray(np.concatenate([np.asarray(m1, np.float64).reshape(-1, 3), np.asarray(m2, np.float64).reshape(-1, 3)], 1))
    poses = np.zeros((4, 8))
    ok = load().och_homography_decompose(np.ascontiguousarray(H, np.float64).reshape(9), rays if len(rays) else np.zeros((1, 6)), len(rays), poses)
    return bool(ok), poses


def image_to_3d(px, model10):
    px = np.ascontiguousarray(px, np.float64).reshape(-1, 2)
    rays = np.zeros((len(px), 3))
    if len(px):
        load().och_image_to_3d(px, len(px), np.ascontiguousarray(model10, np.float64), rays)
    return rays


def subsample(loc, strength, spacing, count=0):
    loc = np.ascontiguousarray(loc, np.float64)
    strength = np.ascontiguousarray(strength, np.float32)
    out = np.zeros(max(len(strength), 1), np.uint64)
    n = load().och_subsample(loc, strength, len(strength), spacing, count, out)
    return out[:n].copy()


def matches_from_device(raw, idx1, idx2):
    """raw: MATCH_DTYPE rows of one pair (len == len(idx1))."""
    raw = np.ascontiguousarray(raw, capi.MATCH_DTYPE)
    idx1 = np.ascontiguousarray(idx1, np.uint64)
    idx2 = np.ascontiguousarray(idx2, np.uint64)
    n = max(len(idx1), 1)
    i1, i2, d = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.float64)
    if len(raw) < len(idx1):
        raise ValueError("raw shorter than idx1")
    m = load().och_matches_from_device(raw.ctypes.data, idx1, len(idx1), idx2, len(idx2), i1, i2, d)
    return i1[:m].copy(), i2[:m].copy(), d[:m].copy()


IP_STATS16 = ["step_s", "init_s", "runners_s", "finalize_s", "load_runner_s", "link_runner_s", "relax_runner_s", "features", "sparse_features",
              "images_linked", "images_relaxed", "relax_solves", "relax_iterations", "relax_setup_host_s", "relax_device_s", "images_to_relax_next"]


class InitialProcessing:
    """INITIAL_PROCESSING as the reference pipelines it (src/pipeline/pipeline.cpp:522-570): step(batch) loads the batch, links
    the batch before and relaxes the batch before that, side by side; step() without images drains."""

    def __init__(self, graph, ctx):
        self.g, self.ctx = graph, ctx
        self.h = graph.L.och_initial_processing_create(graph.h, ctx.h)
        if not self.h:
            raise MemoryError("och_initial_processing_create")

    def close(self):
        if self.h:
            self.g.L.och_initial_processing_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    @property
    def pending(self):
        return bool(self.g.L.och_initial_processing_pending(self.h))

    def step(self, images_bgr=None, model=0, positions=None, max_keypoints=30000, device_shape=None, sequential=False):
        if images_bgr is None:
            n, h, w, src, on_dev, pos, ids = 0, 0, 0, None, 0, None, None
        else:
            if device_shape is None:
                imgs = np.ascontiguousarray(images_bgr, np.uint8)
                n, h, w, _ = imgs.shape
                src, on_dev = imgs.ctypes.data, 0
            else:
                n, h, w = device_shape
                src, on_dev = int(images_bgr), 1
            pos = np.ascontiguousarray(positions, np.float64).reshape(-1, 3)
            ids = np.zeros(max(n, 1), np.uint64)
        stats = np.zeros(16)
        rc = self.g.L.och_initial_processing_step(self.h, src, n, w, h, max_keypoints, on_dev, model,
                                                  None if pos is None else pos.ctypes.data, int(sequential),
                                                  None if ids is None else ids.ctypes.data, stats)
        if rc != 0:
            raise capi.OchipError("initial processing step failed: " + self.g.L.och_last_error(self.g.h).decode())
        if n:
            self.g.node_ids += [int(i) for i in ids[:n]]
        return dict(zip(IP_STATS16, stats.tolist()))
