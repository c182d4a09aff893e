"""Fixtures of the reference's relax tests, restated (data only):
test/test_relax.cpp:19-167 (3 cameras + 100 planar points) and :685-812 (grid of cameras over a plane)."""
import numpy as np


def axis_angle(axis, ang):
    axis = np.asarray(axis, float)
    return np.array([*(axis * np.sin(ang / 2)), np.cos(ang / 2)])


def qmul(a, b):  # xyzw, Eigen product a*b
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def qinv(q):
    return np.array([-q[0], -q[1], -q[2], q[3]]) / np.dot(q, q)


def qrot(q, v):
    return qmul(qmul(q, np.array([*v, 0.0])), qinv(q))[:3]


def qangle(a, b):
    """Eigen::AngleAxisd(a.inverse() * b).angle()"""
    d = qmul(qinv(a), b)
    d = d / np.linalg.norm(d)
    return 2 * np.arctan2(np.linalg.norm(d[:3]), abs(d[3]))


DOWN = axis_angle([1, 0, 0], np.pi)
MODEL_600 = np.array([600.0, 400, 300, 0, 0, 0, 0, 0, 800, 600])


def project(q, pos, p, model):
    ray = qrot(qinv(q), (p - pos) / np.linalg.norm(p - pos))
    z = max(ray[2], 1e-3)
    return ray[:2] / z * model[0] + model[1:3]


def three_cameras():
    """init_cameras(): test_relax.cpp:30-59"""
    ori = [qmul(axis_angle([0, 0, 1], 0.2), DOWN), qmul(axis_angle([0, 1, 0], -0.3), DOWN),
           qmul(axis_angle([1, 0, 0], -0.3), DOWN)]
    pos = [np.array([9.0, 9, 9]), np.array([11.0, 9, 9]), np.array([11.0, 11, 9])]
    return np.array(ori), np.array(pos)


def planar_points():
    """generate_planar_points(): test_relax.cpp:61-73"""
    return np.array([[i + 5, j + 5, -10 + 1e-3 * i + 1e-2 * j] for i in range(10) for j in range(10)], float)


def ring_edges(ori, pos, points, model=MODEL_600):
    """add_point_measurements(): edge i -> (i+1)%3, every point an inlier (test_relax.cpp:91-125)"""
    n = len(ori)
    px = [np.array([project(ori[i], pos[i], p, model) for p in points]) for i in range(n)]
    edges = []
    for i in range(n):
        j = (i + 1) % n
        edges.append(dict(src=i, dst=j, H=None, px=np.concatenate([px[i], px[j]], axis=1),
                          match_index=np.arange(len(points)), dist=None))
    return edges


def add_ori_noise(ori, noise):
    """add_ori_noise(): right-multiplied per-camera perturbations (test_relax.cpp:151-156)"""
    axes = [[0, 1, 0], [0, 0, 1], [1, 0, 0]]
    return np.array([qmul(ori[i], axis_angle(axes[i], noise[i])) for i in range(3)])


def camera_grid(rows, cols, seed=3, height=10.0, spacing=2.0, yaw_sigma=0.05, pts_per_side=14):
    """A rows x cols nadir grid over the tilted plane z = 1e-3 x + 1e-2 y with every 4-neighbour pair
    linked in both directions (pattern of test/test_relax.cpp:685-812)."""
    rng = np.random.default_rng(seed)
    n = rows * cols
    pos = np.array([[c * spacing + rng.uniform(-0.1, 0.1), r * spacing + rng.uniform(-0.1, 0.1), height]
                    for r in range(rows) for c in range(cols)])
    ori = np.array([qmul(axis_angle([0, 0, 1], rng.normal(0, yaw_sigma)), DOWN) for _ in range(n)])
    gx = np.linspace(-spacing, cols * spacing, pts_per_side)
    gy = np.linspace(-spacing, rows * spacing, pts_per_side)
    pts = np.array([[x + rng.uniform(-0.2, 0.2), y + rng.uniform(-0.2, 0.2), 0.0] for x in gx for y in gy])
    pts[:, 2] = 1e-3 * pts[:, 0] + 1e-2 * pts[:, 1]
    model = MODEL_600
    px = [np.array([project(ori[i], pos[i], p, model) for p in pts]) for i in range(n)]
    vis = [np.all((px[i] >= 0) & (px[i] < model[8:10]), axis=1) for i in range(n)]
    edges = []
    for r in range(rows):
        for c in range(cols):
            i = r * cols + c
            for dr, dc in ((0, 1), (1, 0), (0, -1), (-1, 0)):
                rr, cc = r + dr, c + dc
                if 0 <= rr < rows and 0 <= cc < cols:
                    j = rr * cols + cc
                    both = np.flatnonzero(vis[i] & vis[j])
                    if len(both) >= 8:
                        edges.append(dict(src=i, dst=j, H=None, px=np.concatenate([px[i][both], px[j][both]], axis=1),
                                          match_index=np.arange(len(both)), dist=None, point_ids=both))
    return ori, pos, edges, model


def features_of_edges(n_nodes, edges):
    """Per node its feature list and per edge the feature indices of its inliers: every inlier of an edge becomes a
    feature of both images (feature indices = positions in the images' feature lists, as assembleInliers records them);
    an inlier that carries 'pid' (a ground point id) shares its feature across the edges of an image, which is what
    makes multi-image tracks (test_relax.cpp:91-125 indexes features by point that way)."""
    feats = [[] for _ in range(n_nodes)]
    index = [dict() for _ in range(n_nodes)]
    plan = []
    for e in edges:
        k = len(e["px"])
        pid = e.get("pid")
        f1, f2 = np.zeros(k, np.uint64), np.zeros(k, np.uint64)
        for j in range(k):
            for node, col, out in ((e["src"], 0, f1), (e["dst"], 2, f2)):
                key = None if pid is None else int(pid[j])
                if key is not None and key in index[node]:
                    out[j] = index[node][key]
                else:
                    out[j] = len(feats[node])
                    if key is not None:
                        index[node][key] = out[j]
                    feats[node].append(e["px"][j][col:col + 2])
        plan.append((f1, f2))
    return [np.array(f, np.float64).reshape(-1, 2) for f in feats], plan


def rx_graph_from_edges(oracle, node_pos, node_ori, model10, edges, paths=None):
    """The flat edge dicts of this module as a MeasurementGraph of oracle/relax_full.cpp."""
    g = oracle.RxGraph()
    g.add_model(model10, 42)
    n = len(node_pos)
    feats, plan = features_of_edges(n, edges)
    for i in range(n):
        g.add_node(node_pos[i], node_ori[i], 0, feats[i], None if paths is None else paths[i])
    for e, (f1, f2) in zip(edges, plan):
        g.add_edge(e["src"], e["dst"], e["px"], f1, f2, e["match_index"], e.get("H"), e.get("dist"))
    return g, feats


def pack_edges_with_features(oracle, n_nodes, edges):
    """pack_edges + 'feat' (inliers x 2 feature indices) and the per-node feature lists, for the product's relax()."""
    feats, plan = features_of_edges(n_nodes, edges)
    pk = oracle.pack_edges(edges)
    total = int(pk["inl_off"][-1])
    feat = np.zeros((max(total, 1), 2), np.uint64)
    o = 0
    for f1, f2 in plan:
        feat[o:o + len(f1), 0], feat[o:o + len(f1), 1] = f1, f2
        o += len(f1)
    pk["feat"] = feat
    return pk, feats


def camera_grid_tracks(rows, cols, **kw):
    """camera_grid whose inliers carry the id of their ground point ('pid'), so that features are shared between the
    edges of an image and multi-image tracks form."""
    ori, pos, edges, model = camera_grid(rows, cols, **kw)
    for e in edges:
        e["pid"] = e["point_ids"]
    return ori, pos, edges, model


def ring_edges_tracks(ori, pos, points, model=MODEL_600):
    """ring_edges with the point id of every inlier (features shared per point, as in test_relax.cpp:91-125)."""
    edges = ring_edges(ori, pos, points, model)
    for e in edges:
        e["pid"] = np.arange(len(points))
    return edges


def grid_5x5():
    """incremental_relax fixture (test_relax.cpp:685-812): 5 x 5 cameras, 20 x 20 planar points, edges between cameras
    closer than 3.5 (i < j), features appended per edge (no sharing)."""
    n = 25
    pos = np.array([[10 + (i % 5) * 2, 10 + (i // 5) * 2, 10.0] for i in range(n)])
    ori = np.array([qmul(axis_angle([0, 0, 1], 0.05 * (i - n / 2.0)), DOWN) for i in range(n)])
    pts = np.array([[9 + i * 0.5, 9 + j * 0.5, -5 + 1e-3 * i + 1e-2 * j] for i in range(20) for j in range(20)])
    px = [np.array([project(ori[i], pos[i], p, MODEL_600) for p in pts]) for i in range(n)]
    vis = [np.all((px[i] >= 0) & (px[i] < MODEL_600[8:10]), axis=1) for i in range(n)]
    edges = []
    for i in range(n):
        for j in range(i + 1, n):
            if np.linalg.norm(pos[i] - pos[j]) < 3.5:
                both = np.flatnonzero(vis[i] & vis[j])
                if len(both):
                    edges.append(dict(src=i, dst=j, H=None, px=np.concatenate([px[i][both], px[j][both]], axis=1),
                                      match_index=both, dist=None))
    return ori, pos, edges


def host_graph_from_edges(host, node_pos, node_ori, model10, edges):
    """The flat edge dicts as an och_graph of the host library (descriptor-less features: locations only)."""
    n = len(node_pos)
    feats, plan = features_of_edges(n, edges)
    g = host.Graph()
    m = g.add_model(model10)
    for i in range(n):
        k = len(feats[i])
        g.add_image(feats[i] if k else np.zeros((0, 2)), np.zeros(k, np.float32), np.zeros((k, 8), np.uint64), k, m, node_pos[i])
    g.set_orientations(node_ori)
    for e, (f1, f2) in zip(edges, plan):
        g.add_edge(g.node_ids[e["src"]], g.node_ids[e["dst"]], e["px"], f1, f2, e["match_index"], e.get("H"), e.get("dist"))
    return g


def host_paths(n):
    return ["synthetic_%d" % i for i in range(n)]  # the paths och_graph_add_image gives its images
