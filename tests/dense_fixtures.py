"""A small survey with dense features for the dense guided matching tests: nadir cameras over a gently rolling ground,
every ground point observed by the cameras that see it, per image a block of sparse features first (random, they must be
ignored) and the dense ones after it (projected ground points with pixel noise and descriptor bit flips, plus distractors)."""
import numpy as np

from opencalibration_amd import synth


def dense_scene(rows=3, cols=4, width=1000, height_px=750, focal=750.0, n_points=2500, n_sparse=40, flips=15, pixel_sigma=0.3,
                distractors=60, seed=21, distortion=None, cam_height=100.0):
    rng = np.random.default_rng(seed)
    model = np.array([focal, width / 2, height_px / 2, 0, 0, 0, 0, 0, width, height_px], np.float64)
    if distortion is not None:
        model[3:8] = distortion
    n = rows * cols
    foot_w, foot_h = cam_height * width / focal, cam_height * height_px / focal
    r_idx, c_idx = np.divmod(np.arange(n), cols)
    pos = np.stack([c_idx * 0.35 * foot_w + rng.uniform(-1, 1, n), r_idx * 0.45 * foot_h + rng.uniform(-1, 1, n),
                    cam_height + rng.uniform(-2, 2, n)], -1)
    yaw = rng.normal(0, 0.2, n)
    tilt = rng.normal(0, 0.03, (n, 2))
    down = np.array([1.0, 0, 0, 0])
    q = np.stack([tilt[:, 0] / 2, tilt[:, 1] / 2, np.sin(yaw / 2), np.cos(yaw / 2)], -1)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    ori = synth.quat_mul(q, down[None, :])
    ground = lambda x, y: 1e-3 * x + 1e-2 * y + 1.5 * np.sin(x / 40.0) * np.cos(y / 55.0)
    gx = rng.uniform(pos[:, 0].min() - foot_w / 2, pos[:, 0].max() + foot_w / 2, n_points)
    gy = rng.uniform(pos[:, 1].min() - foot_h / 2, pos[:, 1].max() + foot_h / 2, n_points)
    P = np.stack([gx, gy, ground(gx, gy)], -1)
    pdesc = synth.descriptors_for_ids(np.arange(n_points, dtype=np.int64) + 1000 * seed)
    feats, num_sparse, point_of = [], [], []
    for i in range(n):
        R = synth.quat_to_matrix(ori[i])
        ray = (P - pos[i]) @ R
        xn = ray[:, :2] / ray[:, 2:3]
        if distortion is not None:
            r2 = np.sum(xn * xn, 1, keepdims=True)
            k1, k2, k3, p1, p2 = distortion
            rad = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
            xy = xn[:, :1] * xn[:, 1:2]
            t = np.array([p1, p2])
            xn = rad * xn + 2 * xy * t[None, :] + t[None, ::-1] * (r2 + 2 * xn * xn)
        px = focal * xn + model[1:3] + rng.normal(0, pixel_sigma, (n_points, 2))
        vis = (ray[:, 2] > 0) & (px[:, 0] >= 0) & (px[:, 0] < width) & (px[:, 1] >= 0) & (px[:, 1] < height_px)
        ids = np.nonzero(vis)[0]
        d = pdesc[ids].copy()
        for r in range(len(ids)):                      # per-observation bit flips (test/test_dense.cpp:97-105)
            for b in rng.choice(486, flips, replace=False):
                d[r, b >> 6] ^= np.uint64(1) << np.uint64(b & 63)
        s_loc = np.stack([rng.uniform(0, width, n_sparse), rng.uniform(0, height_px, n_sparse)], -1)
        s_desc = synth.descriptors_for_ids(rng.integers(1 << 40, 1 << 41, n_sparse))
        x_loc = np.stack([rng.uniform(0, width, distractors), rng.uniform(0, height_px, distractors)], -1)
        x_desc = synth.descriptors_for_ids(rng.integers(1 << 41, 1 << 42, distractors))
        order = rng.permutation(len(ids) + distractors)
        dl = np.concatenate([px[ids], x_loc])[order]
        dd = np.concatenate([d, x_desc])[order]
        pid = np.concatenate([ids, -np.ones(distractors, np.int64)])[order]
        loc = np.concatenate([s_loc, dl])
        desc = np.concatenate([s_desc, dd])
        feats.append((np.ascontiguousarray(loc), np.ascontiguousarray(desc)))
        num_sparse.append(n_sparse)
        point_of.append(pid)
    return dict(model=model, position=pos, orientation=ori, features=feats, num_sparse=np.array(num_sparse, np.uint64),
                points=P, point_of=point_of, ground=ground)


def ground_mesh_arrays(host_or_oracle_rebuild, scene):
    """The surface the matching runs against: rebuildMesh over the cameras with the vertices put on the true ground."""
    s = host_or_oracle_rebuild(scene["position"])
    a = s.arrays()
    v = a["vertices"].copy()
    v[:, 2] = scene["ground"](v[:, 0], v[:, 1])
    return v, a["edges"]


def host_graph(host, scene):
    g = host.Graph()
    m = g.add_model(scene["model"])
    for i, (loc, desc) in enumerate(scene["features"]):
        st = np.linspace(1.0, 0.1, len(loc)).astype(np.float32)
        g.add_image(loc, st, desc, int(scene["num_sparse"][i]), m, scene["position"][i])
    g.set_orientations(scene["orientation"])
    return g
