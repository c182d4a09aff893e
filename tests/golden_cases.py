"""Cases of tests/golden/oracle_r01.json (written by scripts/make_golden.py): what is computed, and how the
oracle and the device path each produce the same record."""
import hashlib

import numpy as np

LINK_CASES = [dict(rows=2, cols=3, feats=512, seed=21, distortion=None),
              dict(rows=2, cols=3, feats=512, seed=33, distortion=(-0.05, 0.01, -0.002, 1e-3, -5e-4))]
EXTRACT_CASES = [dict(w=320, h=240, seed=1), dict(w=500, h=333, seed=3), dict(w=800, h=600, seed=11)]


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def pair_record(a, b, i1, i2, dist, inliers, n_inliers, score, H, accepted):
    return {"a": int(a), "b": int(b), "matches": digest(np.asarray(i1, np.uint64), np.asarray(i2, np.uint64), np.asarray(dist, np.float64)),
            "inliers": digest(np.asarray(inliers, np.uint8)), "n_matches": int(len(i1)), "n_inliers": int(n_inliers),
            "score": float(score).hex(), "H": [float(v).hex() for v in np.asarray(H).ravel()], "accepted": bool(accepted)}


def knn_order(grid, a):
    xy = grid.position[:, :2]
    d2 = ((xy - xy[a]) ** 2).sum(1)
    return [int(b) for b in np.argsort(d2, kind="stable")[:10] if b != a]


def oracle_link_case(pyoracle, synth, case):
    grid = synth.make_grid(case["rows"], case["cols"], feats=case["feats"], seed=case["seed"], distortion=case["distortion"])
    subsets = [pyoracle.subsample(*grid.image(i)[:2], 40.0, int(grid.num_sparse[i])) for i in range(grid.n_images)]
    pairs = []
    for a in range(grid.n_images):
        for b in knn_order(grid, a):
            la, _, da, _ = grid.image(a)
            lb, _, db, _ = grid.image(b)
            e = pyoracle.link_pair(la, da, subsets[a], lb, db, subsets[b], grid.model, grid.model)
            pairs.append(pair_record(a, b, e["i1"], e["i2"], e["dist"], e["inliers"], e["n_inliers"], e["score"], e["H"],
                                     e["accepted"]))
    return dict(case, distortion=list(case["distortion"]) if case["distortion"] else None, subsets=digest(*subsets), pairs=pairs)


def device_link_case(ctx, host, synth, case):
    grid = synth.make_grid(case["rows"], case["cols"], feats=case["feats"], seed=case["seed"], distortion=case["distortion"])
    g = host.Graph.from_synthetic(grid)
    g.link(ctx, keep_debug=True)
    index_of = {nid: i for i, nid in enumerate(g.node_ids)}
    subsets = [host.subsample(*grid.image(i)[:2], 40.0, int(grid.num_sparse[i])) for i in range(grid.n_images)]
    edges = {(index_of[e["source"]], index_of[e["dest"]]): e for e in g.edges()}
    pairs = []
    for d in g.link_debug():
        a, b = index_of[d["node"]], index_of[d["match_node"]]
        e = edges[(a, b)]
        pairs.append(pair_record(a, b, d["i1"], d["i2"], d["dist"], d["inliers"], int(np.sum(d["inliers"])), d["score"], e["H"],
                                 e["n_matches"] > 0))
    g.close()
    order = {(a, b): k for k, (a, b) in enumerate((a, b) for a in range(grid.n_images) for b in knn_order(grid, a))}
    pairs.sort(key=lambda r: order[(r["a"], r["b"])])
    return dict(case, distortion=list(case["distortion"]) if case["distortion"] else None, subsets=digest(*subsets), pairs=pairs)


def _extract_record(case, kp, desc, loc, st, d, ns):
    order = np.lexsort((kp[:, 0], kp[:, 1], kp[:, 5]))
    return dict(case, n_keypoints=int(len(kp)), keypoints=digest(kp[order].view(np.uint32)), descriptors=digest(desc[order]),
                n_features=int(len(st)), num_sparse=int(ns), features=digest(loc, st, d))


def oracle_extract_case(pyoracle, synth, case):
    img = synth.render_blobs(case["w"], case["h"], case["seed"])
    kp, desc = pyoracle.akaze(img[:, :, 0])
    return _extract_record(case, kp, desc, *pyoracle.extract_features(img))


def device_extract_case(ctx, host, synth, case):
    img = synth.render_blobs(case["w"], case["h"], case["seed"])
    (kp, desc), = ctx.akaze_batch(img[None], max_kp=30000)[0]
    return _extract_record(case, kp, desc, *host.extract_features_batch(ctx, img[None])[0])
