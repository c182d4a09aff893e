"""Deterministic synthetic aerial grids.

Stands in for the image files the reference loads: it produces, per image, exactly what
`extract_features` hands to the link stage — `feature_2d` records (pixel location, strength,
486-bit descriptor; include/opencalibration/types/feature_2d.hpp:9-21) — plus the GPS position
and camera model of `image` (types/image.hpp:17-33).  Pattern follows the reference's own
synthetic fixtures (test/test_dense.cpp:82-105: descriptor = PRNG(point id) with a few bit flips
per observation; test/test_relax.cpp:697-812: nadir grid over a tilted plane).

Data only: no algorithm of the hot path lives here.
"""
from dataclasses import dataclass, field

import numpy as np

DESCRIPTOR_BITS = 486
_LAST_WORD_MASK = np.uint64((1 << (DESCRIPTOR_BITS - 448)) - 1)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def descriptors_for_ids(ids):
    """486-bit descriptor of a ground point, a pure function of its id (8 little-endian u64 words,
    bits 486..511 zero, the std::bitset<486> layout of SURVEY.md a1)."""
    ids = np.asarray(ids, np.uint64)
    with np.errstate(over="ignore"):
        w = _splitmix64(ids[:, None] * np.uint64(8) + np.arange(8, dtype=np.uint64)[None, :])
    w[:, 7] &= _LAST_WORD_MASK
    return np.ascontiguousarray(w)


def quat_mul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], -1)


def quat_to_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


@dataclass
class SyntheticGrid:
    n_images: int
    model: np.ndarray            # [f, ppx, ppy, k1, k2, k3, p1, p2, cols, rows]
    position: np.ndarray         # (N,3) GPS position, local metres
    orientation: np.ndarray      # (N,4) true camera->world quaternion, xyzw
    off: np.ndarray              # (N+1,) feature offsets
    loc: np.ndarray              # (F,2) f64 pixel locations
    strength: np.ndarray         # (F,) f32
    desc: np.ndarray             # (F,8) u64
    point_id: np.ndarray         # (F,) i64, -1 for distractors
    num_sparse: np.ndarray       # (N,) u64
    plane: tuple = (1e-3, 1e-2)  # ground z = a*x + b*y
    meta: dict = field(default_factory=dict)

    def image(self, i):
        s = slice(int(self.off[i]), int(self.off[i + 1]))
        return self.loc[s], self.strength[s], self.desc[s], self.point_id[s]


def make_grid(rows, cols, feats=4096, seed=12345, distractor_frac=0.3, flips=20, pixel_sigma=0.5, yaw_sigma=0.05,
              height=100.0, along=25.0, cross=45.0, distortion=None, mismatch_frac=0.0):
    """rows x cols lawn-mower grid.  `feats` = ground points per footprint (they sit on a jittered
    lattice >= 40 px apart in every image, so they all survive the 40 px subsample of
    link_stage.cpp:63-65).  `mismatch_frac`: share of an image's ground points that keep their descriptor but sit at a
    random pixel - they still match across images and are true outliers for RANSAC."""
    rng = np.random.Generator(np.random.PCG64(seed))
    W, Hh, f = 4000, 3000, 3000.0
    model = np.array([f, W / 2, Hh / 2, 0, 0, 0, 0, 0, W, Hh], np.float64)
    if distortion is not None:
        model[3:8] = distortion
    pa, pb = 1e-3, 1e-2

    n = rows * cols
    r_idx, c_idx = np.divmod(np.arange(n), cols)
    c_eff = np.where(r_idx % 2 == 0, c_idx, cols - 1 - c_idx)  # lawn-mower: alternate direction per strip
    pos = np.zeros((n, 3))
    pos[:, 0] = c_eff * along + rng.uniform(-0.5, 0.5, n)      # jitter: no kNN distance ties
    pos[:, 1] = r_idx * cross + rng.uniform(-0.5, 0.5, n)
    pos[:, 2] = height + pa * pos[:, 0] + pb * pos[:, 1]       # constant height above the tilted ground

    yaw = rng.normal(0, yaw_sigma, n)
    down = np.array([1.0, 0, 0, 0])                            # AngleAxis(pi, X), relax.cpp:12
    qyaw = np.stack([np.zeros(n), np.zeros(n), np.sin(yaw / 2), np.cos(yaw / 2)], -1)
    quat = quat_mul(qyaw, down[None, :])

    # ground lattice: spacing so that one footprint holds ~feats points
    foot_w, foot_h = height * W / f, height * Hh / f
    s = np.sqrt(foot_w * foot_h / feats)
    margin = 0.75 * np.hypot(foot_w, foot_h)
    x0, y0 = -margin, -margin
    nx = int(np.ceil(((cols - 1) * along + 2 * margin) / s)) + 1
    ny = int(np.ceil(((rows - 1) * cross + 2 * margin) / s)) + 1
    gx = x0 + (np.arange(nx)[None, :] + rng.uniform(-0.06, 0.06, (ny, nx))) * s
    gy = y0 + (np.arange(ny)[:, None] + rng.uniform(-0.06, 0.06, (ny, nx))) * s
    gz = pa * gx + pb * gy
    gid = (np.arange(ny)[:, None] * nx + np.arange(nx)[None, :]).astype(np.int64)

    locs, strs, descs, pids, offs = [], [], [], [], [0]
    half = 0.5 * np.hypot(foot_w, foot_h) * 1.1
    for i in range(n):
        R = quat_to_matrix(quat[i])
        ix0, ix1 = max(int((pos[i, 0] - half - x0) / s), 0), min(int((pos[i, 0] + half - x0) / s) + 2, nx)
        iy0, iy1 = max(int((pos[i, 1] - half - y0) / s), 0), min(int((pos[i, 1] + half - y0) / s) + 2, ny)
        P = np.stack([gx[iy0:iy1, ix0:ix1].ravel(), gy[iy0:iy1, ix0:ix1].ravel(), gz[iy0:iy1, ix0:ix1].ravel()], -1)
        ids = gid[iy0:iy1, ix0:ix1].ravel()
        ray = (P - pos[i]) @ R                                  # R^T (P - C)
        px = f * ray[:, :2] / ray[:, 2:3] + model[1:3]
        if distortion is not None:
            xn = ray[:, :2] / ray[:, 2:3]
            r2 = np.sum(xn * xn, 1, keepdims=True)
            k1, k2, k3, p1, p2 = distortion
            rad = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
            xy = xn[:, :1] * xn[:, 1:2]
            t = np.array([p1, p2])
            xd = rad * xn + 2 * xy * t[None, :] + t[None, ::-1] * (r2 + 2 * xn * xn)
            px = f * xd + model[1:3]
        px = px + rng.normal(0, pixel_sigma, px.shape)
        if mismatch_frac > 0:   # (no draw otherwise: the default grids keep their random stream)
            moved = rng.uniform(0, 1, len(px)) < mismatch_frac
            px[moved] = np.stack([rng.uniform(0, W, int(moved.sum())), rng.uniform(0, Hh, int(moved.sum()))], -1)
        ok = (px[:, 0] >= 0) & (px[:, 0] < W) & (px[:, 1] >= 0) & (px[:, 1] < Hh) & (ray[:, 2] > 0)
        px, ids = px[ok], ids[ok]
        d = descriptors_for_ids(ids)
        if flips > 0 and len(ids):
            bits = rng.integers(0, DESCRIPTOR_BITS, (len(ids), flips))
            for j in range(flips):
                b = bits[:, j]
                d[np.arange(len(ids)), b >> 6] ^= np.uint64(1) << (b & 63).astype(np.uint64)
        nd = int(round(distractor_frac * len(ids)))
        dpx = np.stack([rng.uniform(0, W, nd), rng.uniform(0, Hh, nd)], -1)
        dd = rng.integers(0, 2 ** 63, (nd, 8), dtype=np.int64).astype(np.uint64) * np.uint64(2) \
            + rng.integers(0, 2, (nd, 8)).astype(np.uint64)
        dd[:, 7] &= _LAST_WORD_MASK
        loc = np.concatenate([px, dpx])
        des = np.concatenate([d, dd])
        pid = np.concatenate([ids, -np.ones(nd, np.int64)])
        st = (1.0 - rng.uniform(0, 1, len(loc))).astype(np.float32)   # (0,1]
        order = np.argsort(-st, kind="stable")                 # extract_features.cpp:55-56: strongest first
        locs.append(loc[order]); strs.append(st[order]); descs.append(des[order]); pids.append(pid[order])
        offs.append(offs[-1] + len(loc))

    off = np.array(offs, np.uint64)
    return SyntheticGrid(
        n_images=n, model=model, position=pos, orientation=quat, off=off,
        loc=np.ascontiguousarray(np.concatenate(locs)), strength=np.ascontiguousarray(np.concatenate(strs)),
        desc=np.ascontiguousarray(np.concatenate(descs)), point_id=np.concatenate(pids),
        num_sparse=np.diff(off).astype(np.uint64), plane=(pa, pb),
        meta=dict(rows=rows, cols=cols, feats=feats, seed=seed, lattice_spacing_m=float(s)))


def render_blobs(width, height, seed, n_blobs=None, shift=(0.0, 0.0), rot=0.0, channels=3):
    """Synthetic 8-bit image for the extract stage: a fixed random field of Gaussian blobs (both polarities,
    sigma 2..6 px) on mid-grey, viewed through a planar rigid motion (shift, rot) so that two renderings of one
    seed are two views of the same scene.  Data only."""
    rng = np.random.default_rng(seed)
    n = n_blobs if n_blobs is not None else max(32, int(width * height / 768))
    px = rng.uniform(-50, width + 50, n)
    py = rng.uniform(-50, height + 50, n)
    amp = rng.uniform(0.2, 0.8, n) * rng.choice([-1.0, 1.0], n)
    sg = rng.uniform(2.0, 6.0, n)
    c, s = np.cos(rot), np.sin(rot)
    cx, cy = width / 2.0, height / 2.0
    img = np.full((height, width), 0.5)
    for i in range(n):
        # scene position of the blob in this view: invert p = R (q - c) + c + shift
        qx = c * (px[i] - cx - shift[0]) + s * (py[i] - cy - shift[1]) + cx
        qy = -s * (px[i] - cx - shift[0]) + c * (py[i] - cy - shift[1]) + cy
        r = int(np.ceil(4 * sg[i]))
        x0, x1 = max(int(qx) - r, 0), min(int(qx) + r + 1, width)
        y0, y1 = max(int(qy) - r, 0), min(int(qy) + r + 1, height)
        if x0 >= x1 or y0 >= y1:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1]
        img[y0:y1, x0:x1] += amp[i] * np.exp(-((xx - qx) ** 2 + (yy - qy) ** 2) / (2 * sg[i] ** 2))
    g = np.clip(img * 255.0, 0, 255).astype(np.uint8)
    if channels == 1:
        return g
    return np.ascontiguousarray(np.stack([g, g, g], axis=-1))


CONFIGS = {
    "C1": dict(rows=2, cols=5, feats=2048),
    "C2": dict(rows=10, cols=20, feats=4096),
    "C3": dict(rows=25, cols=40, feats=4096),
    "C5": dict(rows=50, cols=100, feats=4096),
}
