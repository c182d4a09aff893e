"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle.so (the CPU restatement of the reference hot path) and of
oracle/_ref/libref.so (the reference's own std-only KD-tree header compiled in place).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing under
opencalibration_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def build():
    """Compile the restatement (and oracle/_ref when /root/reference is mounted)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all", "ref"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oc_libstdcxx_selfcheck.argtypes = [u64p]
        L.oc_subsample.restype = C.c_size_t
        L.oc_subsample.argtypes = [f64p, f32p, C.c_size_t, C.c_double, C.c_size_t, u64p]
        L.oc_match.restype = C.c_size_t
        L.oc_match.argtypes = [u64p, C.c_size_t, u64p, C.c_size_t, u64p, C.c_size_t, u64p, C.c_size_t, u64p, u64p, f64p]
        L.oc_image_to_3d.argtypes = [f64p, C.c_size_t, f64p, f64p]
        L.oc_image_from_3d.argtypes = [f64p, C.c_size_t, f64p, f64p]
        L.oc_homography_fit4.argtypes = [f64p, C.c_size_t, u64p, f64p, f64p]
        L.oc_homography_fit_inliers.argtypes = [f64p, C.c_size_t, u8p, f64p, f64p]
        L.oc_homography_evaluate.restype = C.c_double
        L.oc_homography_evaluate.argtypes = [f64p, C.c_size_t, f64p, f64p, u8p, C.c_void_p]
        L.oc_ransac_homography.restype = C.c_double
        L.oc_ransac_homography.argtypes = [f64p, C.c_size_t, f64p, u8p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.oc_homography_decompose.restype = C.c_int
        L.oc_homography_decompose.argtypes = [f64p, f64p, C.c_size_t, u8p, f64p]
        L.oc_link_pair.argtypes = [f64p, u64p, C.c_size_t, u64p, C.c_size_t, f64p, u64p, C.c_size_t, u64p, C.c_size_t,
                                   f64p, f64p, u64p, u64p, f64p, u8p, f64p, f64p, f64p]
        L.oc_extract_tail.restype = C.c_size_t
        L.oc_extract_tail.argtypes = [f32p, u64p, C.c_size_t, C.c_double, f64p, f32p, u64p, u64p]
        L.oc_refit_edge.argtypes = [f64p, C.c_size_t, f64p, C.c_size_t, f64p, f64p, u64p, u64p, f64p, C.c_size_t, u8p, f64p,
                                    f64p, f64p]
        L.oc_link_batch_cpu.argtypes = [f64p, f32p, u64p, u64p, C.c_size_t, u64p, f64p, u32p, C.c_size_t, C.c_int,
                                        C.c_int, u64p, f64p, f64p]
        L.oc_scene_homography.argtypes = [C.c_size_t, C.c_size_t, C.c_uint, f64p, u8p, f64p]
        L.oc_scene_near_degenerate.argtypes = [f64p, f64p]
        L.oc_num_threads.restype = C.c_int
        vp = C.c_void_p
        L.oc_relax_ground_plane.argtypes = [C.c_size_t, f64p, f64p, f64p, C.c_size_t, u64p, f64p, C.c_size_t, u64p, u64p,
                                            f64p, u8p, u64p, f64p, u64p, vp, vp, C.c_size_t, u64p, f64p, f64p]
        L.oc_points_downwards_prior.restype = C.c_double
        L.oc_points_downwards_prior.argtypes = [f64p, C.c_double]
        L.oc_robust_centroid.argtypes = [f64p, C.c_int, C.c_double, f64p]
        L.oc_plane_intersection_cost.restype = C.c_int
        L.oc_plane_intersection_cost.argtypes = [f64p, f64p, f64p, f64p, f64p, f64p, f64p, vp]
        f32p_ = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        L.oc_akaze.restype = C.c_size_t
        L.oc_akaze.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, f32p_, u64p, vp]
        L.oc_akaze_level.restype = C.c_size_t
        L.oc_akaze_level.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, f32p_, vp, vp]
        L.oc_extract_features.restype = C.c_size_t
        L.oc_extract_features.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, f64p, f32p_, u64p, u64p]
        L.oc_gray_resize.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
        _LIB = L
    return _LIB


def akaze(gray, max_kp=50000):
    """Restated AKAZE on an 8-bit grey image: (kp6 [x, y, size, angle, response, level], desc [n x 8])."""
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    kp = np.zeros((max_kp, 6), np.float32)
    d = np.zeros((max_kp, 8), np.uint64)
    n = lib().oc_akaze(gray, w, h, max_kp, kp, d, None)
    assert n <= max_kp
    return kp[:n].copy(), d[:n].copy()


def extract_features(bgr, max_n=100000):
    """extract_features(cv::Mat) restated: (loc [n x 2] f64 full-res pixels, strength, desc, num_sparse)."""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    loc, st, d = np.zeros((max_n, 2)), np.zeros(max_n, np.float32), np.zeros((max_n, 8), np.uint64)
    ns = np.zeros(1, np.uint64)
    n = lib().oc_extract_features(bgr, w, h, max_n, loc, st, d, ns)
    assert n <= max_n
    return loc[:n].copy(), st[:n].copy(), d[:n].copy(), int(ns[0])


def extract_tail(kp6, desc, scale):
    """The tail of extract_features (sort by strength, 8 px NMS) on given keypoints in detection order."""
    kp6 = np.ascontiguousarray(kp6, np.float32).reshape(-1, 6)
    desc = np.ascontiguousarray(desc, np.uint64).reshape(-1, 8)
    n = len(kp6)
    loc, st, d, ns = np.zeros((n + 1, 2)), np.zeros(n + 1, np.float32), np.zeros((n + 1, 8), np.uint64), np.zeros(1, np.uint64)  # the seed keypoint appears twice
    m = lib().oc_extract_tail(kp6 if n else np.zeros((1, 6), np.float32), desc if n else np.zeros((1, 8), np.uint64), n, float(scale), loc, st, d, ns)
    return loc[:m].copy(), st[:m].copy(), d[:m].copy(), int(ns[0])


def gray_resize(bgr, ow, oh):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    out = np.zeros((oh, ow), np.uint8)
    lib().oc_gray_resize(bgr, w, h, out, ow, oh)
    return out


def pack_edges(edges):
    """edges: list of dicts {src, dst, H (3x3) or None, px (k x 4), match_index (k,), dist (m,) or None}.
    Returns the flat arrays oc_relax_ground_plane (and the product's relax entry points) take."""
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


def relax_ground_plane(node_pos, node_ori, model10, pose_node, pose_ori, edges, opt_edges=None):
    """relax(graph, nodes, cam_models, edges, {ORIENTATION, GROUND_PLANE}) of the restatement."""
    node_pos = np.ascontiguousarray(node_pos, np.float64)
    node_ori = np.ascontiguousarray(node_ori, np.float64)
    pose_node = np.ascontiguousarray(pose_node, np.uint64)
    pose_ori = np.ascontiguousarray(pose_ori, np.float64).copy()
    pk = pack_edges(edges)
    opt = np.ascontiguousarray(np.arange(len(edges)) if opt_edges is None else opt_edges, np.uint64)
    plane, summary = np.zeros(9), np.zeros(6)
    lib().oc_relax_ground_plane(len(node_pos), node_pos, node_ori, np.ascontiguousarray(model10, np.float64),
                                len(pose_node), pose_node, pose_ori, len(edges), pk["src"], pk["dst"], pk["H"],
                                pk["is_h"], pk["inl_off"], pk["px"], pk["match_index"], pk["dist_off"].ctypes.data,
                                pk["dist"].ctypes.data, len(opt), opt, plane, summary)
    return dict(orientation=pose_ori, plane=plane.reshape(3, 3), solves=int(summary[0]),
                iterations_total=int(summary[1]), last_iterations=int(summary[2]), initial_cost=summary[3],
                final_cost=summary[4], residual_blocks=int(summary[5]))


def points_downwards_prior(q, weight):
    return lib().oc_points_downwards_prior(np.ascontiguousarray(q, np.float64), weight)


def robust_centroid(points, thr):
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    out = np.zeros(3)
    lib().oc_robust_centroid(points, len(points), thr, out)
    return out


def plane_intersection_cost(locs, rays, plane_xy, q0, q1, z, want_jac=True):
    res, jac = np.zeros(6), np.zeros((6, 11))
    c = lambda a: np.ascontiguousarray(a, np.float64)
    ok = lib().oc_plane_intersection_cost(c(locs), c(rays), c(plane_xy), c(q0), c(q1), c(z), res,
                                          jac.ctypes.data if want_jac else None)
    return bool(ok), res, jac


def ref():
    """oracle/_ref/libref.so or None if it was never built (it needs /root/reference to build)."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_subsample_kdtree.restype = C.c_size_t
        R.ref_subsample_kdtree.argtypes = [f64p, f32p, C.c_size_t, C.c_double, C.c_size_t, u64p]
        R.ref_knn.argtypes = [f64p, C.c_size_t, C.c_size_t, u64p]
        _REF = R
    return _REF


def model_vec(f, ppx, ppy, radial=(0, 0, 0), tangential=(0, 0), cols=4000, rows=3000):
    return np.array([f, ppx, ppy, *radial, *tangential, cols, rows], dtype=np.float64)


def selfcheck():
    out = np.zeros(17, np.uint64)
    lib().oc_libstdcxx_selfcheck(out)
    return out


def subsample(loc, strength, spacing, count=0):
    loc = np.ascontiguousarray(loc, np.float64)
    strength = np.ascontiguousarray(strength, np.float32)
    out = np.zeros(max(len(strength), 1), np.uint64)
    n = lib().oc_subsample(loc, strength, len(strength), spacing, count, out)
    return out[:n].copy()


def ref_subsample(loc, strength, spacing, count=0):
    loc = np.ascontiguousarray(loc, np.float64)
    strength = np.ascontiguousarray(strength, np.float32)
    out = np.zeros(max(len(strength), 1), np.uint64)
    n = ref().ref_subsample_kdtree(loc, strength, len(strength), spacing, count, out)
    return out[:n].copy()


def ref_knn(xy, k):
    xy = np.ascontiguousarray(xy, np.float64)
    out = np.zeros((len(xy), k), np.uint64)
    ref().ref_knn(xy, len(xy), k, out)
    return out


def match(desc1, desc2, idx1, idx2):
    desc1 = np.ascontiguousarray(desc1, np.uint64)
    desc2 = np.ascontiguousarray(desc2, np.uint64)
    idx1 = np.ascontiguousarray(idx1, np.uint64)
    idx2 = np.ascontiguousarray(idx2, np.uint64)
    n = max(len(idx1), 1)
    i1, i2, d = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.float64)
    m = lib().oc_match(desc1, len(desc1), desc2, len(desc2), idx1, len(idx1), idx2, len(idx2), i1, i2, d)
    return i1[:m].copy(), i2[:m].copy(), d[:m].copy()


def image_to_3d(px, model):
    px = np.ascontiguousarray(px, np.float64).reshape(-1, 2)
    out = np.zeros((len(px), 3))
    lib().oc_image_to_3d(px, len(px), model, out)
    return out


def image_from_3d(rays, model):
    rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 3)
    out = np.zeros((len(rays), 2))
    lib().oc_image_from_3d(rays, len(rays), model, out)
    return out


def corr_array(m1, m2, quality=None):
    m1 = np.asarray(m1, np.float64).reshape(-1, 3)
    m2 = np.asarray(m2, np.float64).reshape(-1, 3)
    q = np.zeros(len(m1)) if quality is None else np.asarray(quality, np.float64)
    return np.ascontiguousarray(np.concatenate([m1, m2, q[:, None]], axis=1))


def fit4(corr, idx4):
    H, Hi = np.zeros((3, 3)), np.zeros((3, 3))
    lib().oc_homography_fit4(corr, len(corr), np.asarray(idx4, np.uint64), H, Hi)
    return H, Hi


def fit_inliers(corr, inl):
    H, Hi = np.zeros((3, 3)), np.zeros((3, 3))
    lib().oc_homography_fit_inliers(corr, len(corr), np.ascontiguousarray(inl, np.uint8), H, Hi)
    return H, Hi


def evaluate(corr, H, Hi):
    inl = np.zeros(max(len(corr), 1), np.uint8)
    err = np.zeros(max(len(corr), 1), np.float64)
    s = lib().oc_homography_evaluate(corr, len(corr), np.ascontiguousarray(H), np.ascontiguousarray(Hi), inl,
                                     err.ctypes.data)
    return s, inl[:len(corr)], err[:len(corr)]


def ransac_homography(corr, max_trace=0):
    corr = np.ascontiguousarray(corr, np.float64).reshape(-1, 7)
    M = len(corr)
    H = np.zeros((3, 3))
    inl = np.zeros(max(M, 1), np.uint8)
    it = np.zeros(2, np.uint64)
    tr = np.zeros((max(max_trace, 1), 4), np.uint64)
    if M == 0:
        corr = np.zeros((1, 7))
    s = lib().oc_ransac_homography(corr, M, H, inl, tr.ctypes.data if max_trace else None, max_trace, it.ctypes.data)
    return dict(score=s, H=H, inliers=inl[:M].copy(), iterations=int(it[0]), improvements=int(it[1]),
                samples=tr[:min(max_trace, int(it[0]))].copy())


def decompose(H, corr, inl):
    poses = np.zeros((4, 8))
    ok = lib().oc_homography_decompose(np.ascontiguousarray(H, np.float64), corr, len(corr),
                                       np.ascontiguousarray(inl, np.uint8), poses)
    return bool(ok), poses


def link_pair(loc1, desc1, idx1, loc2, desc2, idx2, model1, model2):
    loc1 = np.ascontiguousarray(loc1, np.float64)
    loc2 = np.ascontiguousarray(loc2, np.float64)
    desc1 = np.ascontiguousarray(desc1, np.uint64)
    desc2 = np.ascontiguousarray(desc2, np.uint64)
    idx1 = np.ascontiguousarray(idx1, np.uint64)
    idx2 = np.ascontiguousarray(idx2, np.uint64)
    n = max(len(idx1), 1)
    mi1, mi2, md = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n)
    inl = np.zeros(n, np.uint8)
    H, poses, summary = np.zeros((3, 3)), np.zeros((4, 8)), np.zeros(7)
    lib().oc_link_pair(loc1, desc1, len(desc1), idx1, len(idx1), loc2, desc2, len(desc2), idx2, len(idx2), model1,
                       model2, mi1, mi2, md, inl, H, poses, summary)
    m = int(summary[0])
    return dict(i1=mi1[:m].copy(), i2=mi2[:m].copy(), dist=md[:m].copy(), inliers=inl[:m].copy(), H=H, poses=poses,
                n_inliers=int(summary[1]), can_decompose=bool(summary[2]), accepted=bool(summary[3]),
                score=float(summary[4]), iterations=int(summary[5]), improvements=int(summary[6]))


def refit_edge(loc1, loc2, model1, model2, i1, i2, dist, inliers):
    """relax_group.cpp:142-176 for one edge: matches (i1, i2, dist) with their previous inlier flags."""
    loc1 = np.ascontiguousarray(loc1, np.float64)
    loc2 = np.ascontiguousarray(loc2, np.float64)
    i1 = np.ascontiguousarray(i1, np.uint64)
    i2 = np.ascontiguousarray(i2, np.uint64)
    dist = np.ascontiguousarray(dist, np.float64)
    M = len(i1)
    inl = np.zeros(max(M, 1), np.uint8)
    inl[:M] = inliers
    H, poses, summary = np.zeros((3, 3)), np.zeros((4, 8)), np.zeros(3)
    pad = lambda a, dt: a if M else np.zeros(1, dt)
    lib().oc_refit_edge(loc1, len(loc1), loc2, len(loc2), np.ascontiguousarray(model1, np.float64),
                        np.ascontiguousarray(model2, np.float64), pad(i1, np.uint64), pad(i2, np.uint64), pad(dist, np.float64), M,
                        inl, H, poses, summary)
    return dict(H=H, inliers=inl[:M].copy(), poses=poses, n_inliers=int(summary[0]), can_decompose=bool(summary[1]),
                accepted=bool(summary[2]))


def scene_homography(n_in, n_out, seed):
    n = n_in + n_out
    corr, gt, H = np.zeros((n, 7)), np.zeros(n, np.uint8), np.zeros((3, 3))
    lib().oc_scene_homography(n_in, n_out, seed, corr, gt, H)
    return corr, gt, H


def scene_fundamental(n_in, n_out, planar_fraction, seed):
    """SyntheticScene::fundamental of test/test_ransac_benchmark.cpp:60-122."""
    n = n_in + n_out
    corr, gt, F = np.zeros((n, 7)), np.zeros(n, np.uint8), np.zeros((3, 3))
    L = lib()
    L.oc_scene_fundamental.argtypes = [C.c_size_t, C.c_size_t, C.c_double, C.c_uint, f64p, u8p, f64p]
    L.oc_scene_fundamental(n_in, n_out, float(planar_fraction), seed, corr, gt, F)
    return corr, gt, F


# ---- a9: fundamental / essential matrix models (oracle/epipolar.cpp); rays: n x 6 {measurement1, measurement2}
def _epipolar_lib():
    L = lib()
    L.oc_ransac_epipolar.restype = C.c_double
    L.oc_ransac_epipolar.argtypes = [C.c_int, f64p, C.c_void_p, C.c_size_t, C.c_double, f64p, u8p, u64p]
    L.oc_epipolar_fit_inliers.argtypes = [C.c_int, f64p, C.c_size_t, u8p, f64p]
    L.oc_epipolar_evaluate.restype = C.c_double
    L.oc_epipolar_evaluate.argtypes = [f64p, f64p, C.c_size_t, C.c_double, u8p, f64p]
    L.oc_essential_decompose.restype = C.c_int
    L.oc_essential_decompose.argtypes = [f64p, f64p]
    L.oc_jacobi_svd.argtypes = [f64p, C.c_int, f64p, f64p, f64p]
    return L


def ransac_epipolar(model, rays, quality=None, threshold=0.0):
    """ransac<fundamental_matrix_model> (model 0) / ransac<essential_matrix_model> (model 1), ransac.cpp:53-257.
    Returns (score, matrix 3x3, inliers, iterations)."""
    rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
    n = len(rays)
    M, inl, it = np.zeros((3, 3)), np.zeros(max(n, 1), np.uint8), np.zeros(1, np.uint64)
    q = None if quality is None else np.ascontiguousarray(quality, np.float64)
    score = _epipolar_lib().oc_ransac_epipolar(model, rays if n else np.zeros((1, 6)), None if q is None else q.ctypes.data, n,
                                               float(threshold), M, inl, it)
    return score, M, inl[:n].astype(bool), int(it[0])


def epipolar_fit_inliers(model, rays, inliers):
    rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
    M = np.zeros((3, 3))
    _epipolar_lib().oc_epipolar_fit_inliers(model, rays, len(rays), np.ascontiguousarray(inliers, np.uint8), M)
    return M


def epipolar_evaluate(M, rays, threshold=0.0):
    """(score, inliers, Sampson errors) of fundamental_matrix_model::evaluate / ::error."""
    rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
    inl, err = np.zeros(len(rays), np.uint8), np.zeros(len(rays))
    score = _epipolar_lib().oc_epipolar_evaluate(np.ascontiguousarray(M, np.float64), rays, len(rays), float(threshold), inl, err)
    return score, inl.astype(bool), err


def essential_decompose(E):
    poses = np.zeros((4, 7))
    ok = _epipolar_lib().oc_essential_decompose(np.ascontiguousarray(E, np.float64), poses)
    return bool(ok), poses


def jacobi_svd(A):
    A = np.ascontiguousarray(A, np.float64)
    n = A.shape[0]
    U, S, V = np.zeros((n, n)), np.zeros(n), np.zeros((n, n))
    _epipolar_lib().oc_jacobi_svd(A, n, U, S, V)
    return U, S, V


def scene_near_degenerate():
    corr, H = np.zeros((100, 7)), np.zeros((3, 3))
    lib().oc_scene_near_degenerate(corr, H)
    return corr, H


# ---------------------------------------------------------------------------------------------------------------
# oracle/relax_full.cpp: the whole relax stage over an in-memory MeasurementGraph (all flavours, RelaxGroup, meshes)
OPT = dict(ORIENTATION=1 << 0, POSITION=1 << 1, GROUND_PLANE=1 << 2, GROUND_MESH=1 << 3, POINTS_3D=1 << 4,
           FOCAL_LENGTH=1 << 5, PRINCIPAL_POINT=1 << 6, LENS_DISTORTIONS_RADIAL=1 << 7, BROWN2=1 << 8, BROWN24=1 << 9,
           BROWN246=1 << 10, LENS_DISTORTIONS_TANGENTIAL=1 << 11, MINIMAL_MESH=1 << 12)
_RX = False


def _rx():
    global _RX
    L = lib()
    if not _RX:
        vp, sz = C.c_void_p, C.c_size_t
        L.ocx_graph_create.restype = vp
        L.ocx_graph_destroy.argtypes = [vp]
        L.ocx_graph_add_model.restype = sz
        L.ocx_graph_add_model.argtypes = [vp, f64p, sz]
        L.ocx_graph_get_model.argtypes = [vp, sz, f64p]
        L.ocx_graph_set_model.argtypes = [vp, sz, f64p]
        L.ocx_graph_add_node.restype = sz
        L.ocx_graph_add_node.argtypes = [vp, C.c_char_p, f64p, f64p, sz, sz, f64p]
        L.ocx_graph_add_edge.restype = sz
        L.ocx_graph_add_edge.argtypes = [vp, sz, sz, vp, C.c_int, sz, f64p, u64p, sz, f64p, vp]
        L.ocx_graph_set_orientation.argtypes = [vp, sz, f64p]
        L.ocx_graph_get_orientations.argtypes = [vp, f64p]
        L.ocx_surface_create.restype = vp
        L.ocx_surface_destroy.argtypes = [vp]
        for f in (L.ocx_surface_num_vertices, L.ocx_surface_num_edges, L.ocx_surface_num_cloud_points):
            f.restype = sz
            f.argtypes = [vp]
        L.ocx_surface_get.argtypes = [vp, vp, vp, vp]
        L.ocx_surface_set.argtypes = [vp, sz, f64p, sz, u64p, sz, f64p]
        L.ocx_rebuild_mesh.argtypes = [f64p, sz, vp, C.c_int, vp]
        L.ocx_surface_triangle_at.restype = C.c_int
        L.ocx_surface_triangle_at.argtypes = [vp, C.c_double, C.c_double, C.c_double, u64p, u64p]
        L.ocx_relax.restype = C.c_int
        L.ocx_relax.argtypes = [vp, sz, u64p, f64p, sz, u64p, C.c_uint32, C.c_double, vp, vp, f64p, vp, sz, vp, sz]
        L.ocx_relax_group.restype = sz
        L.ocx_relax_group.argtypes = [vp, sz, u64p, u64p, sz, C.c_uint32, C.c_double, vp, vp, C.c_int, f64p, u64p, sz, u64p,
                                      sz, vp]
        L.ocx_convert_model.argtypes = [f64p, C.c_int, f64p]
        L.ocx_image_to_3d_inverse.argtypes = [f64p, sz, f64p, f64p]
        _RX = True
    return L


def options(*names):
    bits = 0
    for n in names:
        bits |= OPT[n]
    return bits


class RxSurface:
    """surface_model (mesh + point cloud) of the restatement."""

    def __init__(self):
        self.h = _rx().ocx_surface_create()

    def __del__(self):
        if getattr(self, "h", None):
            _rx().ocx_surface_destroy(self.h)
            self.h = None

    def arrays(self):
        L = _rx()
        nv, ne, nc = L.ocx_surface_num_vertices(self.h), L.ocx_surface_num_edges(self.h), L.ocx_surface_num_cloud_points(self.h)
        v, e, c = np.zeros((max(nv, 1), 3)), np.zeros((max(ne, 1), 5), np.uint64), np.zeros((max(nc, 1), 3))
        L.ocx_surface_get(self.h, v.ctypes.data, e.ctypes.data, c.ctypes.data)
        return dict(vertices=v[:nv], edges=e[:ne], cloud=c[:nc])

    def set(self, vertices, edges, cloud=None):
        v = np.ascontiguousarray(vertices, np.float64).reshape(-1, 3)
        e = np.ascontiguousarray(edges, np.uint64).reshape(-1, 5)
        c = np.zeros((0, 3)) if cloud is None else np.ascontiguousarray(cloud, np.float64).reshape(-1, 3)
        _rx().ocx_surface_set(self.h, len(v), v if len(v) else np.zeros((1, 3)), len(e), e if len(e) else np.zeros((1, 5), np.uint64),
                              len(c), c if len(c) else np.zeros((1, 3)))
        return self

    def triangle_at(self, x, y, z_from=1e3):
        tri, steps = np.zeros(3, np.uint64), np.zeros(1, np.uint64)
        t = _rx().ocx_surface_triangle_at(self.h, x, y, z_from, tri, steps)
        return t, tri, int(steps[0])


def densify_mesh(positions, orientations, models10, features, num_sparse, surface, match_cap=1 << 22):
    """densifyMesh restated (oracle/dense.cpp).  features: per image (loc k x 2, desc k x 8 u64); models10: one row per
    image.  The merged points are appended to `surface` (an RxSurface) and returned with the accepted matches."""
    L = _rx()
    n = len(features)
    off = np.concatenate([[0], np.cumsum([len(f[0]) for f in features])]).astype(np.uint64)
    loc = np.ascontiguousarray(np.concatenate([np.asarray(f[0], np.float64).reshape(-1, 2) for f in features]))
    desc = np.ascontiguousarray(np.concatenate([np.asarray(f[1], np.uint64).reshape(-1, 8) for f in features]))
    pairs = np.zeros((match_cap, 2), np.uint64)
    points = np.zeros((int(off[-1]) + 1, 3))
    counts = np.zeros(3, np.uint64)
    L.ocx_densify.argtypes = [C.c_size_t, f64p, f64p, f64p, u64p, f64p, u64p, u64p, C.c_void_p, C.c_void_p, C.c_size_t,
                              C.c_void_p, C.c_size_t, u64p]
    L.ocx_densify.restype = None
    L.ocx_densify(n, np.ascontiguousarray(positions, np.float64), np.ascontiguousarray(orientations, np.float64),
                  np.ascontiguousarray(models10, np.float64).reshape(n, 10), off, loc, desc,
                  np.ascontiguousarray(num_sparse, np.uint64), surface.h, pairs.ctypes.data, match_cap, points.ctypes.data,
                  len(points), counts)
    return dict(matches=int(counts[0]), tracks=int(counts[1]), points=points[:int(counts[2])].copy(),
                match_pairs=pairs[:min(int(counts[0]), match_cap)].copy())


def hilbert_xy2d(order, x, y):
    L = lib()
    L.ocx_hilbert_xy2d.restype = C.c_uint32
    return int(L.ocx_hilbert_xy2d(int(order), int(x), int(y)))


def rebuild_mesh(cam_xyz, prev=None, minimal=False):
    cam_xyz = np.ascontiguousarray(cam_xyz, np.float64).reshape(-1, 3)
    s = RxSurface()
    _rx().ocx_rebuild_mesh(cam_xyz, len(cam_xyz), prev.h if prev is not None else None, int(minimal), s.h)
    return s


class RxGraph:
    """MeasurementGraph of the restatement (node / edge ids = insertion indices)."""

    def __init__(self):
        self.h = _rx().ocx_graph_create()
        self.n_nodes = 0
        self.n_edges = 0

    def __del__(self):
        if getattr(self, "h", None):
            _rx().ocx_graph_destroy(self.h)
            self.h = None

    def add_model(self, model10, model_id=42):
        return _rx().ocx_graph_add_model(self.h, np.ascontiguousarray(model10, np.float64), model_id)

    def get_model(self, idx):
        m = np.zeros(10)
        _rx().ocx_graph_get_model(self.h, idx, m)
        return m

    def set_model(self, idx, model10):
        _rx().ocx_graph_set_model(self.h, idx, np.ascontiguousarray(model10, np.float64))

    def add_node(self, pos, ori, model_index=0, features=None, path=None):
        f = np.zeros((0, 2)) if features is None else np.ascontiguousarray(features, np.float64).reshape(-1, 2)
        path = ("%08d" % self.n_nodes) if path is None else path
        i = _rx().ocx_graph_add_node(self.h, path.encode(), np.ascontiguousarray(pos, np.float64),
                                     np.ascontiguousarray(ori, np.float64), model_index, len(f), f if len(f) else np.zeros((1, 2)))
        self.n_nodes += 1
        return i

    def add_edge(self, src, dst, px, fidx1, fidx2, match_index=None, H=None, dist=None, poses=None):
        px = np.ascontiguousarray(px, np.float64).reshape(-1, 4)
        n = len(px)
        idx = np.zeros((max(n, 1), 3), np.uint64)
        idx[:n, 0], idx[:n, 1] = fidx1, fidx2
        idx[:n, 2] = np.arange(n) if match_index is None else match_index
        Hc = None if H is None else np.ascontiguousarray(H, np.float64)
        d = np.zeros(0) if dist is None else np.ascontiguousarray(dist, np.float64)
        pc = None if poses is None else np.ascontiguousarray(poses, np.float64)
        e = _rx().ocx_graph_add_edge(self.h, src, dst, Hc.ctypes.data if Hc is not None else None, int(H is not None), n,
                                     px if n else np.zeros((1, 4)), idx, len(d), d if len(d) else np.zeros(1),
                                     pc.ctypes.data if pc is not None else None)
        self.n_edges += 1
        return e

    def persist_cam_models(self, on=True):
        """Keep the cam_models map of relax() across calls (the graph's own models stay untouched), as a caller of the
        reference's relax() does when it passes the same map again."""
        L = _rx()
        L.ocx_graph_persist_cam_models.argtypes = [C.c_void_p, C.c_int]
        L.ocx_graph_persist_cam_models(self.h, int(on))

    def set_orientation(self, node, q):
        _rx().ocx_graph_set_orientation(self.h, node, np.ascontiguousarray(q, np.float64))

    def orientations(self):
        o = np.zeros((self.n_nodes, 4))
        _rx().ocx_graph_get_orientations(self.h, o)
        return o

    def relax(self, pose_node, pose_ori, opt_edges, opts, grid_fraction=0.1, prev=None):
        pose_node = np.ascontiguousarray(pose_node, np.uint64)
        pose_ori = np.ascontiguousarray(pose_ori, np.float64).copy()
        opt_edges = np.ascontiguousarray(opt_edges, np.uint64)
        out = RxSurface()
        summary, iters, models = np.zeros(9), np.full(64, -1, np.int32), np.zeros((8, 11))
        nm = _rx().ocx_relax(self.h, len(pose_node), pose_node, pose_ori, len(opt_edges),
                             opt_edges if len(opt_edges) else np.zeros(1, np.uint64), opts, grid_fraction,
                             prev.h if prev is not None else None, out.h, summary, iters.ctypes.data, len(iters),
                             models.ctypes.data, len(models))
        return dict(orientation=pose_ori, surface=out, solves=int(summary[0]), iterations_total=int(summary[1]),
                    last_iterations=int(summary[2]), initial_cost=summary[3], final_cost=summary[4],
                    residual_blocks=int(summary[5]), parameter_blocks=int(summary[6]), track_blocks=int(summary[7]),
                    two_ray_blocks=int(summary[8]), iterations_per_solve=[int(i) for i in iters if i >= 0],
                    models={int(m[0]): m[1:].copy() for m in models[:nm]})

    def points_problem(self, pose_node, pose_ori, opt_edges, opts, mode, model10=None):
        """TestRelaxProblem of test/test_relax.cpp:470-483 on the restatement: setup3dPointProblem, then nothing (mode 0),
        solve (1) or relaxObservedModelOnly (2).  Returns the tracks' points before / after, the summary of the last solve,
        the orientations and the camera model afterwards."""
        L = _rx()
        pose_node = np.ascontiguousarray(pose_node, np.uint64)
        pose_ori = np.ascontiguousarray(pose_ori, np.float64).copy()
        opt_edges = np.ascontiguousarray(opt_edges, np.uint64)
        cap = 100000
        before, after, summary = np.zeros((cap, 3)), np.zeros((cap, 3)), np.zeros(9)
        m = None if model10 is None else np.ascontiguousarray(model10, np.float64).copy()
        L.ocx_points_problem.restype = C.c_size_t
        L.ocx_points_problem.argtypes = [C.c_void_p, C.c_size_t, u64p, f64p, C.c_size_t, u64p, C.c_uint32, C.c_int, f64p, f64p,
                                         C.c_size_t, f64p, C.c_void_p]
        n = L.ocx_points_problem(self.h, len(pose_node), pose_node, pose_ori, len(opt_edges), opt_edges, opts, mode, before, after,
                                 cap, summary, None if m is None else m.ctypes.data)
        return dict(points_before=before[:n].copy(), points_after=after[:n].copy(), orientation=pose_ori, solves=int(summary[0]),
                    iterations=int(summary[2]), iterations_total=int(summary[1]), initial_cost=summary[3], final_cost=summary[4],
                    residual_blocks=int(summary[5]), model=m)

    def relax_group(self, node_ids, knn10, depth, opts, grid_fraction=0.1, prev=None, run=True):
        node_ids = np.ascontiguousarray(node_ids, np.uint64)
        knn10 = np.ascontiguousarray(knn10, np.uint64).reshape(self.n_nodes, 10)
        out = RxSurface()
        summary = np.zeros(9)
        cap = 4 * self.n_nodes + 8
        local, edges, ne = np.zeros(cap, np.uint64), np.zeros(self.n_edges + 1, np.uint64), C.c_size_t(0)
        n = _rx().ocx_relax_group(self.h, len(node_ids), node_ids, knn10, depth, opts, grid_fraction,
                                  prev.h if prev is not None else None, out.h, int(run), summary, local, cap, edges,
                                  len(edges), C.byref(ne))
        return dict(local_nodes=local[:n].copy(), opt_edges=edges[:ne.value].copy(), surface=out, solves=int(summary[0]),
                    iterations_total=int(summary[1]), residual_blocks=int(summary[5]))


def convert_model(model10, to_inverse):
    out = np.zeros(10)
    _rx().ocx_convert_model(np.ascontiguousarray(model10, np.float64), int(to_inverse), out)
    return out


def image_to_3d_inverse(px, inverse_model10):
    px = np.ascontiguousarray(px, np.float64).reshape(-1, 2)
    out = np.zeros((len(px), 3))
    _rx().ocx_image_to_3d_inverse(px, len(px), np.ascontiguousarray(inverse_model10, np.float64), out)
    return out


def knn10_bruteforce(xy):
    """imageGPSLocations.searchKnn(position, 10) (self included) by exhaustive search, ties to the lower index."""
    xy = np.asarray(xy, np.float64)
    n = len(xy)
    out = np.full((n, 10), np.iinfo(np.uint64).max, np.uint64)
    for i in range(n):
        d = np.sum((xy - xy[i]) ** 2, axis=1)
        order = np.argsort(d, kind="stable")[:10]
        out[i, :len(order)] = order
    return out


def relax_stage_groups(rxgraph, node_ids=None, disable_parallelism=False, opts=0, ordered=False):
    """RelaxStage::init's partition (oracle/relax_cluster.cpp): (number of groups, group of every node or -1, context depth)."""
    L = _rx()
    L.ocx_relax_stage_groups.restype = C.c_size_t
    i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
    L.ocx_relax_stage_groups.argtypes = [C.c_void_p, C.c_size_t, u64p, C.c_int, C.c_int, C.c_uint32, i64p, C.c_void_p, i64p]
    ids = np.zeros(1, np.uint64) if node_ids is None else np.ascontiguousarray(node_ids, np.uint64)
    out = np.full(max(rxgraph.n_nodes, 1), -1, np.int64)
    depth = C.c_size_t(0)
    pos = np.full(max(rxgraph.n_nodes, 1), -1, np.int64)
    n = L.ocx_relax_stage_groups(rxgraph.h, 0 if node_ids is None else len(ids), ids, int(node_ids is None),
                                 int(disable_parallelism), opts, out, C.addressof(depth), pos)
    out, pos = out[:rxgraph.n_nodes].copy(), pos[:rxgraph.n_nodes].copy()
    if ordered:   # the groups as lists of node ids in the order the clustering holds them
        return [[int(i) for i in sorted(np.flatnonzero(out == g), key=lambda i: pos[i])] for g in range(n)], depth.value
    return n, out, depth.value


def kmeans3(xyz, k, iterations, use_ref=False):
    """KMeans<size_t, 3>: the restatement, or (use_ref) the reference's own header compiled into oracle/_ref."""
    xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
    a, c, s = np.zeros(len(xyz), np.uint64), np.zeros((k, 3)), np.zeros(k, np.uint64)
    fn = ref().ref_kmeans3 if use_ref else lib().ocx_kmeans3
    fn.argtypes = [f64p, C.c_size_t, C.c_size_t, C.c_int, u64p, f64p, u64p]
    fn(xyz, len(xyz), k, iterations, a, c, s)
    return a, c, s


class RxMesh:
    """The reference's mesh refinement restated a second time (oracle/refine_mesh.cpp, from src/surface/refine_mesh.cpp): a
    mesh held in an emulation of the reference's graph container, so that its iteration orders - which the refinement's
    choices depend on - persist from one refinement to the next.  Vertices are numbered in creation order."""

    NONE = np.iinfo(np.uint64).max

    def __init__(self, vertices, edges5):
        L = lib()
        L.ocx_rmesh_create.restype = C.c_void_p
        L.ocx_rmesh_create.argtypes = [f64p, C.c_size_t, u64p, C.c_size_t]
        L.ocx_rmesh_destroy.argtypes = [C.c_void_p]
        L.ocx_rmesh_counts.argtypes = [C.c_void_p, u64p, u64p]
        L.ocx_rmesh_get.argtypes = [C.c_void_p, f64p, u64p]
        L.ocx_rmesh_set_heights.argtypes = [C.c_void_p, f64p]
        L.ocx_rmesh_count_points.restype = C.c_size_t
        L.ocx_rmesh_count_points.argtypes = [C.c_void_p, C.c_size_t, u64p, f64p, u64p, f64p, C.c_size_t]
        L.ocx_rmesh_refine_by_point_density.restype = C.c_size_t
        L.ocx_rmesh_refine_by_point_density.argtypes = [C.c_void_p, C.c_size_t, u64p, f64p, C.c_size_t, C.c_double, C.c_int, C.c_double]
        L.ocx_rmesh_refine_at_point.restype = C.c_size_t
        L.ocx_rmesh_refine_at_point.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
        self.L = L
        v = np.ascontiguousarray(vertices, np.float64).reshape(-1, 3)
        e = np.ascontiguousarray(edges5, np.uint64).reshape(-1, 5)
        self.h = L.ocx_rmesh_create(v if len(v) else np.zeros((1, 3)), len(v), e if len(e) else np.zeros((1, 5), np.uint64), len(e))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ocx_rmesh_destroy(self.h)
            self.h = None

    def arrays(self):
        nv, ne = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        self.L.ocx_rmesh_counts(self.h, nv, ne)
        v, e = np.zeros((max(int(nv[0]), 1), 3)), np.zeros((max(int(ne[0]), 1), 5), np.uint64)
        self.L.ocx_rmesh_get(self.h, v, e)
        return v[:int(nv[0])], e[:int(ne[0])]

    def set_heights(self, z):
        self.L.ocx_rmesh_set_heights(self.h, np.ascontiguousarray(z, np.float64))

    @staticmethod
    def _clouds(clouds):
        sizes = np.array([len(c) for c in clouds] or [0], np.uint64)
        xyz = np.ascontiguousarray(np.concatenate([np.asarray(c, np.float64).reshape(-1, 3) for c in clouds])
                                   if len(clouds) and sizes.sum() else np.zeros((1, 3)))
        return len(clouds), sizes, xyz

    def count_points_per_triangle(self, clouds):
        n, sizes, xyz = self._clouds(clouds)
        cap = 2 * max(len(self.arrays()[1]), 1)
        tri, st = np.zeros((cap, 3), np.uint64), np.zeros((cap, 2))
        k = self.L.ocx_rmesh_count_points(self.h, n, sizes, xyz, tri, st, cap)
        return tri[:k], st[:k, 0].astype(np.int64), st[:k, 1]

    def refine_by_point_density(self, clouds, max_points_per_triangle, min_distance_variance=0.0, max_iterations=10, min_triangle_size=0.0):
        n, sizes, xyz = self._clouds(clouds)
        return self.L.ocx_rmesh_refine_by_point_density(self.h, n, sizes, xyz, max_points_per_triangle, min_distance_variance,
                                                        max_iterations, min_triangle_size)

    def refine_at_point(self, x, y, levels=1):
        return self.L.ocx_rmesh_refine_at_point(self.h, x, y, levels)
