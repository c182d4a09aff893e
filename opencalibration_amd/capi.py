"""ctypes binding of libochip.so (include/ochip.h).  No fallbacks: if the library is missing or no
gfx950 device is present the calls raise."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libochip.so")

K_MATCH, K_RANSAC, K_RELAX_EVAL, K_RELAX_SOLVE, K_AKAZE = 0, 1, 2, 3, 4
NO_SECOND = 0xFFFF

PAIR_DTYPE = np.dtype([("image_1", np.uint32), ("image_2", np.uint32)])
MATCH_DTYPE = np.dtype([("best_k", np.uint32), ("best_count", np.uint16), ("second_count", np.uint16)])

# every symbol include/ochip.h declares; tests check that the built library exports all of them
EXPORTS = [
    "ochip_ctx_create", "ochip_ctx_destroy", "ochip_ctx_sibling", "ochip_ctx_set_priority", "ochip_last_error", "ochip_device_info", "ochip_synchronize",
    "ochip_descriptors_reserve", "ochip_upload_descriptors", "ochip_descriptor_count",
    "ochip_match_batch", "ochip_match_launch", "ochip_match_fetch",
    "ochip_upload_keypoints", "ochip_ransac_homography_batch", "ochip_refit_homography_batch", "ochip_ransac_epipolar_batch",
    "ochip_upload_batch", "ochip_host_alloc", "ochip_host_free", "ochip_akaze_batch", "ochip_akaze_batch_dev", "ochip_akaze_features", "ochip_akaze_features_dev", "ochip_feature_lists_from_keypoints",
    "ochip_synth_views_alloc", "ochip_synth_views_free", "ochip_synth_render_views", "ochip_synth_views_read",
    "ochip_relax_problem_create", "ochip_relax_problem_destroy", "ochip_relax_set_cameras_constant",
    "ochip_relax_solve", "ochip_relax_get_state", "ochip_relax_set_shard",
    "ochip_plane_setup_create", "ochip_plane_setup_override", "ochip_plane_setup_blocks", "ochip_plane_setup_destroy",
    "ochip_plane_chain_create", "ochip_plane_chain_run", "ochip_plane_chain_destroy",
    "ochip_relaxg_problem_create", "ochip_relaxg_problem_destroy", "ochip_relaxg_set_structure_only", "ochip_relaxg_solve",
    "ochip_relaxg_get_state", "ochip_relaxg_evaluate", "ochip_relaxg_set_exchange",
    "ochip_relaxp_problem_create", "ochip_relaxp_problem_destroy", "ochip_relaxp_set_structure_only", "ochip_relaxp_solve",
    "ochip_relaxp_get_state",
    "ochip_profile_reset", "ochip_profile_get", "ochip_match_work", "ochip_relax_work", "ochip_relax_memory", "ochip_work_counters",
    "ochip_debug_fp64", "ochip_debug_std_sort", "ochip_match_sort", "ochip_ransac_homography_batch_sorted", "ochip_edge_lists",
    "ochip_dense_index_create", "ochip_dense_index_destroy", "ochip_dense_match", "ochip_dense_link", "ochip_dense_triangulate",
    "ochip_rccl_unique_id", "ochip_rccl_comm_create", "ochip_rccl_comm_destroy", "ochip_rccl_comm_stats",
    "ochip_rccl_relax_exchange",
]

_lib = None


class OchipError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OchipError(f"{LIB_PATH} is missing: run `python -m opencalibration_amd.build` "
                             "(there is no CPU fallback for the hot path)")
        L = C.CDLL(LIB_PATH)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.ochip_ctx_create.argtypes = [i32, C.POINTER(vp)]
        L.ochip_ctx_destroy.argtypes = [vp]
        L.ochip_ctx_destroy.restype = None
        L.ochip_last_error.argtypes = [vp]
        L.ochip_last_error.restype = C.c_char_p
        L.ochip_device_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(i32), C.POINTER(C.c_size_t)]
        L.ochip_synchronize.argtypes = [vp]
        L.ochip_rccl_unique_id.argtypes = [vp, vp]
        L.ochip_rccl_comm_create.argtypes = [vp, vp, u32, u32, C.POINTER(vp)]
        L.ochip_rccl_comm_destroy.argtypes = [vp]
        L.ochip_rccl_comm_destroy.restype = None
        L.ochip_rccl_comm_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        L.ochip_descriptors_reserve.argtypes = [vp, u32, u64]
        L.ochip_upload_descriptors.argtypes = [vp, u32, vp, u32]
        L.ochip_descriptor_count.argtypes = [vp, u32, C.POINTER(u32)]
        L.ochip_match_batch.argtypes = [vp, vp, u32, vp, vp]
        L.ochip_match_launch.argtypes = [vp, vp, u32, vp, u64]
        L.ochip_match_fetch.argtypes = [vp, vp, u64]
        L.ochip_profile_reset.argtypes = [vp]
        L.ochip_profile_get.argtypes = [vp, i32, C.POINTER(u64), C.POINTER(C.c_double)]
        L.ochip_match_work.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        L.ochip_relax_work.argtypes = [vp, C.POINTER(C.c_double)]
        L.ochip_relax_memory.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
        L.ochip_debug_fp64.argtypes = [vp, i32, vp, vp, C.c_size_t, vp]
        L.ochip_debug_std_sort.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp]
        L.ochip_akaze_batch.argtypes = [vp, vp, u32, i32, i32, u32, vp, vp, vp, vp]
        L.ochip_feature_lists_from_keypoints.argtypes = [vp, vp, vp, vp, u32, u32, i32, i32, C.c_double, C.c_double, vp]
        L.ochip_akaze_batch_dev.argtypes = [vp, vp, u32, i32, i32, u32, vp, vp, vp, vp]
        L.ochip_synth_views_alloc.argtypes = [vp, u32, i32, i32, C.POINTER(vp)]
        L.ochip_synth_views_free.argtypes = [vp, vp]
        L.ochip_synth_views_free.restype = None
        L.ochip_synth_render_views.argtypes = [vp, vp, u32, u32, i32, i32, vp, vp, vp, vp, u32]
        L.ochip_synth_views_read.argtypes = [vp, vp, u32, i32, i32, vp]
        _lib = L
    return _lib


class RcclComm:
    """ochip_rccl_comm: all-gathers on the owning context's stream (include/ochip.h)."""

    def __init__(self, ctx, unique_id, rank, world):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self.h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        ctx._check(ctx.L.ochip_rccl_comm_create(ctx.h, buf, self.rank, self.world, C.byref(self.h)), "ochip_rccl_comm_create")

    @property
    def exchange(self):
        """(function pointer, user pointer) for ochip_relax_set_shard."""
        return C.cast(self.ctx.L.ochip_rccl_relax_exchange, C.c_void_p).value, self.h

    def stats(self):
        n, b = C.c_uint64(), C.c_uint64()
        self.ctx._check(self.ctx.L.ochip_rccl_comm_stats(self.h, C.byref(n), C.byref(b)), "ochip_rccl_comm_stats")
        return {"exchanges": n.value, "bytes_gathered": b.value}

    def close(self):
        if self.h:
            self.ctx.L.ochip_rccl_comm_destroy(self.h)
            self.h = C.c_void_p()


class Context:
    """Owns one ochip_ctx (one GPU)."""

    def __init__(self, device=0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.ochip_ctx_create(device, C.byref(h))
        if rc != 0:
            raise OchipError(f"ochip_ctx_create({device}) = {rc}: {self.L.ochip_last_error(None).decode()}")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.L.ochip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise OchipError(f"{what} = {rc}: {self.L.ochip_last_error(self.h).decode()}")

    def sibling(self, index):
        """The index-th sibling context (same device, own streams and scratch; owned by this context): independent work
        submitted through it overlaps with this context's."""
        h = C.c_void_p()
        self._check(self.L.ochip_ctx_sibling(self.h, index, C.byref(h)), "ochip_ctx_sibling")
        s = Context.__new__(Context)
        s.L, s.h, s._owner = self.L, h, self
        s.close = lambda: None          # the owner destroys it
        return s

    def rccl_unique_id(self):
        """ncclGetUniqueId as bytes (rank 0 draws it and hands it to the other ranks)."""
        buf = (C.c_uint8 * 128)()
        self._check(self.L.ochip_rccl_unique_id(self.h, buf), "ochip_rccl_unique_id")
        return bytes(buf)

    def rccl_comm(self, unique_id, rank, world):
        """This rank's RCCL communicator on this context (RcclComm): the native transport of the sharded relax."""
        return RcclComm(self, unique_id, rank, world)

    def set_priority(self, high=True):
        self._check(self.L.ochip_ctx_set_priority(self.h, int(high)), "ochip_ctx_set_priority")

    def device_info(self):
        name = C.create_string_buffer(256)
        cu, mem = C.c_int(), C.c_size_t()
        self._check(self.L.ochip_device_info(self.h, name, 256, C.byref(cu), C.byref(mem)), "ochip_device_info")
        return dict(name=name.value.decode(), compute_units=cu.value, hbm_bytes=mem.value)

    def synchronize(self):
        self._check(self.L.ochip_synchronize(self.h), "ochip_synchronize")

    def descriptors_reserve(self, n_images, total):
        self._check(self.L.ochip_descriptors_reserve(self.h, n_images, total), "ochip_descriptors_reserve")

    def upload_descriptors(self, image_id, desc):
        desc = np.ascontiguousarray(desc, np.uint64).reshape(-1, 8)
        self._check(self.L.ochip_upload_descriptors(self.h, image_id, desc.ctypes.data, len(desc)),
                    "ochip_upload_descriptors")

    def match_batch(self, pairs, out_offset, out_total=None):
        pairs = np.ascontiguousarray(pairs, PAIR_DTYPE)
        out_offset = np.ascontiguousarray(out_offset, np.uint64)
        if out_total is None:
            raise ValueError("out_total required")
        out = np.zeros(max(out_total, 1), MATCH_DTYPE)
        self._check(self.L.ochip_match_launch(self.h, pairs.ctypes.data, len(pairs), out_offset.ctypes.data, out_total),
                    "ochip_match_launch")
        self._check(self.L.ochip_match_fetch(self.h, out.ctypes.data, out_total), "ochip_match_fetch")
        return out[:out_total]

    def match_launch(self, pairs, out_offset, out_total):
        pairs = np.ascontiguousarray(pairs, PAIR_DTYPE)
        out_offset = np.ascontiguousarray(out_offset, np.uint64)
        self._check(self.L.ochip_match_launch(self.h, pairs.ctypes.data, len(pairs), out_offset.ctypes.data, out_total),
                    "ochip_match_launch")

    def akaze_batch(self, images_bgr, max_kp=20000):
        """images_bgr: (n, h, w, 3) uint8.  Returns (list of (kp6, desc) per image, (work_w, work_h))."""
        imgs = np.ascontiguousarray(images_bgr, np.uint8)
        n, h, w, _ = imgs.shape
        kp = np.zeros((n, max_kp, 6), np.float32)
        desc = np.zeros((n, max_kp, 8), np.uint64)
        counts = np.zeros(n, np.uint32)
        wh = np.zeros(2, np.int32)
        self._check(self.L.ochip_akaze_batch(self.h, imgs.ctypes.data, n, w, h, max_kp, kp.ctypes.data, desc.ctypes.data,
                                             counts.ctypes.data, wh.ctypes.data), "ochip_akaze_batch")
        return [(kp[i, :counts[i]].copy(), desc[i, :counts[i]].copy()) for i in range(n)], (int(wh[0]), int(wh[1]))

    def std_sort(self, keys, payload, offsets):
        """ochip_debug_std_sort: (keys, payload, fallback flags) with every segment [offsets[s], offsets[s + 1]) ordered as
        std::sort by descending key leaves it."""
        keys = np.ascontiguousarray(keys, np.uint32)
        payload = np.ascontiguousarray(payload, np.uint32)
        offsets = np.ascontiguousarray(offsets, np.uint32)
        n_segs = len(offsets) - 1
        ko, po = np.zeros(max(len(keys), 1), np.uint32), np.zeros(max(len(keys), 1), np.uint32)
        fb = np.zeros(max(n_segs, 1), np.uint8)
        kin = keys if len(keys) else np.zeros(1, np.uint32)
        pin = payload if len(keys) else np.zeros(1, np.uint32)
        self._check(self.L.ochip_debug_std_sort(self.h, kin.ctypes.data, pin.ctypes.data, offsets.ctypes.data, n_segs, ko.ctypes.data,
                                                po.ctypes.data, fb.ctypes.data), "ochip_debug_std_sort")
        return ko[:len(keys)], po[:len(keys)], fb[:n_segs].astype(bool)

    def feature_lists(self, kp6, desc, work_wh, scale, nms_radius=8.0, subset_spacing=0.0):
        """ochip_feature_lists_from_keypoints for ONE image's keypoints (detection order): dict of records ((n + 1) x 88
        bytes: the output list), response, slot (n each), num_sparse, conflict; with subset_spacing > 0 also subset (indices
        into the feature list) and subset_conflict."""
        kp6 = np.ascontiguousarray(kp6, np.float32).reshape(-1, 6)
        desc = np.ascontiguousarray(desc, np.uint64).reshape(-1, 8)
        n = len(kp6)
        m = max(n, 1)
        rec, resp = np.zeros((m + 1, 88), np.uint8), np.zeros(m, np.float32)
        slot, ns, conflict = np.zeros(m, np.uint32), np.zeros(4, np.uint32), np.zeros(16, np.uint8)
        subset, nsub, sconf = np.zeros(16384, np.uint32), np.zeros(4, np.uint32), np.zeros(16, np.uint8)
        counts = np.array([n], np.uint32)

        class Lists(C.Structure):
            _fields_ = [("records", C.c_void_p), ("response", C.c_void_p), ("slot", C.c_void_p), ("num_sparse", C.c_void_p),
                        ("conflict", C.c_void_p), ("subset", C.c_void_p), ("num_subset", C.c_void_p), ("subset_conflict", C.c_void_p),
                        ("subset_spacing", C.c_double)]

        lists = Lists(rec.ctypes.data, resp.ctypes.data, slot.ctypes.data, ns.ctypes.data, conflict.ctypes.data, subset.ctypes.data,
                      nsub.ctypes.data, sconf.ctypes.data, float(subset_spacing))
        kin = kp6 if n else np.zeros((1, 6), np.float32)
        din = desc if n else np.zeros((1, 8), np.uint64)
        self._check(self.L.ochip_feature_lists_from_keypoints(self.h, kin.ctypes.data, din.ctypes.data, counts.ctypes.data, 1, m,
                                                              int(work_wh[0]), int(work_wh[1]), float(scale), float(nms_radius),
                                                              C.byref(lists)), "ochip_feature_lists_from_keypoints")
        out = dict(records=rec[:n + 1 if n else 0], response=resp[:n], slot=slot[:n], num_sparse=int(ns[0]), conflict=bool(conflict[0]))
        if subset_spacing > 0:
            out.update(subset=subset[:int(nsub[0])].copy(), subset_conflict=bool(sconf[0]))
        return out

    def synth_views(self, position, orientation, width, height, f, pp, plane, spacing, origin, seed=7, chunk=64):
        """Render one synthetic view per camera directly into HBM (benchmark / test data).  Returns an opaque
        device pointer (int) to n x height x width x 3 bytes; free with synth_views_free."""
        n = len(position)
        ptr = C.c_void_p()
        self._check(self.L.ochip_synth_views_alloc(self.h, n, width, height, C.byref(ptr)), "ochip_synth_views_alloc")
        cams = np.ascontiguousarray(np.concatenate([position, orientation], axis=1), np.float64)
        model3 = np.array([f, pp[0], pp[1]], np.float64)
        plane2 = np.array(plane, np.float64)
        lat3 = np.array([origin[0], origin[1], spacing], np.float64)
        for i in range(0, n, chunk):
            m = min(chunk, n - i)
            c = np.ascontiguousarray(cams[i:i + m])
            self._check(self.L.ochip_synth_render_views(self.h, ptr, i, m, width, height, c.ctypes.data, model3.ctypes.data,
                                                        plane2.ctypes.data, lat3.ctypes.data, seed), "ochip_synth_render_views")
        return ptr.value

    def synth_views_read(self, ptr, index, width, height):
        out = np.zeros((height, width, 3), np.uint8)
        self._check(self.L.ochip_synth_views_read(self.h, C.c_void_p(ptr), index, width, height, out.ctypes.data),
                    "ochip_synth_views_read")
        return out

    def synth_views_read_into(self, ptr, index, width, height, out):
        """The same into a caller's (height, width, 3) uint8 array (e.g. a slice of host_array())."""
        self._check(self.L.ochip_synth_views_read(self.h, C.c_void_p(ptr), index, width, height, out.ctypes.data),
                    "ochip_synth_views_read")

    def host_array(self, shape, dtype=np.uint8):
        """A numpy array over page-locked host memory (ochip_host_alloc): PCIe copies from it run at link rate.  Returns
        (array, release); call release() when done (the array must not be used afterwards)."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self.L.ochip_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        self.L.ochip_host_free.argtypes = [C.c_void_p, C.c_void_p]
        self.L.ochip_host_free.restype = None
        self._check(self.L.ochip_host_alloc(self.h, nbytes, C.byref(p)), "ochip_host_alloc")
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        return arr, (lambda: self.L.ochip_host_free(self.h, p))

    def synth_views_free(self, ptr):
        self.L.ochip_synth_views_free(self.h, C.c_void_p(ptr))

    def debug_fp64(self, op, x, y=None):
        x = np.ascontiguousarray(x, np.float64)
        y = np.ascontiguousarray(x if y is None else y, np.float64)
        out = np.zeros_like(x)
        self._check(self.L.ochip_debug_fp64(self.h, op, x.ctypes.data, y.ctypes.data, x.size, out.ctypes.data),
                    "ochip_debug_fp64")
        return out

    def profile_reset(self):
        self._check(self.L.ochip_profile_reset(self.h), "ochip_profile_reset")

    def relax_work(self):
        f = C.c_double()
        self._check(self.L.ochip_relax_work(self.h, C.byref(f)), "ochip_relax_work")
        return f.value

    def work_counters(self):
        """{ransac loop trips x correspondences, relax residual blocks with / without Jacobians} since the last profile reset."""
        out = (C.c_uint64 * 3)()
        self.L.ochip_work_counters.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self.L.ochip_work_counters(self.h, out), "ochip_work_counters")
        return {"ransac_hyp_corr": int(out[0]), "relax_blocks_jac": int(out[1]), "relax_blocks_cost": int(out[2])}

    def relax_memory(self):
        """(unknowns, bytes stored, bytes dense) of the largest reduced system a relax on this context has held."""
        n, b, d = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.L.ochip_relax_memory(self.h, C.byref(n), C.byref(b), C.byref(d)), "ochip_relax_memory")
        return n.value, b.value, d.value

    def match_work(self):
        """(computed, delivered) descriptor distances of the match launches since the last profile_reset."""
        c, d = C.c_uint64(), C.c_uint64()
        self._check(self.L.ochip_match_work(self.h, C.byref(c), C.byref(d)), "ochip_match_work")
        return c.value, d.value

    def profile_get(self, kid):
        n, ms = C.c_uint64(), C.c_double()
        self._check(self.L.ochip_profile_get(self.h, kid, C.byref(n), C.byref(ms)), "ochip_profile_get")
        return n.value, ms.value
