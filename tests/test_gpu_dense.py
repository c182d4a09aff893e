"""Dense guided matching on the device (csrc/dense.hip + host/dense_stereo.cpp) against the restated densifyMesh
(oracle/dense.cpp): the same accepted matches in the same order, the same tracks, the same triangulated points."""
import numpy as np
import pytest

from opencalibration_amd import capi, host
from oracle import pyoracle
from dense_fixtures import dense_scene, ground_mesh_arrays, host_graph

pytestmark = pytest.mark.gpu


def _both(scene, ctx):
    n = len(scene["features"])
    v, e = ground_mesh_arrays(pyoracle.rebuild_mesh, scene)
    exp_surface = pyoracle.RxSurface().set(v, e)
    exp = pyoracle.densify_mesh(scene["position"], scene["orientation"], np.tile(scene["model"], (n, 1)), scene["features"],
                                scene["num_sparse"], exp_surface)
    g = host_graph(host, scene)
    surface = host.Surface().set(v, e)
    got = g.densify_mesh(ctx, surface, want_matches=True)
    return exp, got, surface, exp_surface, g


@pytest.mark.parametrize("kw", [dict(), dict(seed=5, distortion=(0.02, -0.07, 0.1, 1e-4, -2e-4)),
                                dict(seed=8, n_points=700, distractors=5, rows=2, cols=3),
                                dict(seed=3, rows=5, cols=6, n_points=3000),                 # 30 cameras: the 11 nearest are a choice
                                dict(seed=3, rows=5, cols=6, n_points=3000, hook="dense_predict_unstaged")])
def test_densify_matches_the_restatement(kw, monkeypatch):
    kw = dict(kw)
    hook = kw.pop("hook", None)
    if hook:
        monkeypatch.setenv("OCHIP_TEST_HOOKS", hook)
    scene = dense_scene(**kw)
    ctx = capi.Context(0)
    exp, got, surface, exp_surface, g = _both(scene, ctx)
    assert got["matches"] == exp["matches"] > 300
    assert np.array_equal(got["match_pairs"], exp["match_pairs"])               # same matches, same order
    # a track = a component of the matches' union-find with at least two measurements, on both sides
    assert got["tracks"] == exp["tracks"] and got["points"] == len(exp["points"])
    cloud = surface.clouds()[-1]
    assert np.allclose(cloud, exp["points"], rtol=0, atol=1e-9)
    dz = cloud[:, 2] - scene["ground"](cloud[:, 0], cloud[:, 1])
    assert np.median(np.abs(dz)) < 0.2
    g.close(), ctx.close()


@pytest.mark.parametrize("kw", [dict(seed=5, distortion=(0.02, -0.07, 0.1, 1e-4, -2e-4)), dict(seed=3, rows=5, cols=6, n_points=3000)])
def test_track_points_on_the_device_are_the_host_loops(kw, monkeypatch):
    """ochip_dense_triangulate (a thread per track) against the host loop it replaces (OCHIP_TEST_HOOKS=host_triangulation):
    the same tracks get a point and the points are the same doubles - with lens distortion (the iterative inverse) and with
    tracks whose outliers send them to the second intersection."""
    scene = dense_scene(**kw)
    ctx = capi.Context(0)
    v, e = ground_mesh_arrays(pyoracle.rebuild_mesh, scene)
    g = host_graph(host, scene)
    on_device, by_host = host.Surface().set(v, e), host.Surface().set(v, e)
    got_d = g.densify_mesh(ctx, on_device)
    monkeypatch.setenv("OCHIP_TEST_HOOKS", "host_triangulation")
    got_h = g.densify_mesh(ctx, by_host)
    assert got_d["tracks"] == got_h["tracks"] and got_d["points"] == got_h["points"] > 100
    assert got_d["points"] <= got_d["tracks"]
    assert np.array_equal(on_device.clouds()[-1], by_host.clouds()[-1])
    g.close(), ctx.close()


def test_raw_search_against_brute_force():
    """ochip_dense_match alone, through the C ABI: nearest / second nearest Hamming distance and the disc population for
    random queries, against numpy - including empty discs, single-candidate discs, exact ties and the open disc edge."""
    import ctypes as C
    rng = np.random.default_rng(2)
    L = capi.load()
    ctx = capi.Context(0)
    n_img, cell = 3, 151.0
    feats, descs = [], []
    for i in range(n_img):
        k = [900, 1, 300][i]
        loc = rng.uniform(0, 1000, (k, 2)) * [1.0, 0.75]
        d = rng.integers(0, 1 << 62, (k, 8), dtype=np.uint64)
        d[:, 7] &= np.uint64((1 << 38) - 1)
        feats.append(loc), descs.append(d)
    feats[0][5] = feats[0][4] + [150.0, 0.0]                   # exactly radius away from a query placed on feature 4
    descs[0][11] = descs[0][10]                                # two identical descriptors: a tie for the best distance
    feats[0][11] = feats[0][10] + [3.0, 4.0]
    # the index layout: features sorted by cell
    feat_off, cell_off, grid2, origin2, cell_start, sdesc, sloc, perms = [0], [0], [], [], [], [], [], []
    for loc, d in zip(feats, descs):
        ox, oy = np.floor(loc.min(0))
        ncx, ncy = int((loc[:, 0].max() - ox) // cell) + 1, int((loc[:, 1].max() - oy) // cell) + 1
        c = (np.floor((loc[:, 1] - oy) / cell) * ncx + np.floor((loc[:, 0] - ox) / cell)).astype(int)
        perm = np.argsort(c, kind="stable")
        starts = np.searchsorted(c[perm], np.arange(ncx * ncy + 1))
        perms.append(perm), sdesc.append(d[perm]), sloc.append(loc[perm]), cell_start.append(starts.astype(np.uint32))
        grid2 += [ncx, ncy]
        origin2 += [ox, oy]
        feat_off.append(feat_off[-1] + len(loc))
        cell_off.append(cell_off[-1] + ncx * ncy + 1)
    ix = C.c_void_p()
    u64 = lambda a: np.ascontiguousarray(a, np.uint64)
    L.ochip_dense_index_create.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 7 + [C.c_double, C.POINTER(C.c_void_p)]
    keep = [u64(feat_off), np.ascontiguousarray(np.concatenate(sdesc)), np.ascontiguousarray(np.concatenate(sloc)), u64(cell_off),
            np.ascontiguousarray(np.concatenate(cell_start)), np.array(grid2, np.int32), np.array(origin2, np.float64)]
    assert L.ochip_dense_index_create(ctx.h, n_img, *[a.ctypes.data for a in keep], cell, C.byref(ix)) == 0
    qdt = np.dtype([("src", np.uint32), ("cand", np.uint32), ("px", np.float64), ("py", np.float64)])
    rdt = np.dtype([("best", np.uint32), ("bc", np.uint16), ("sc", np.uint16), ("nearby", np.uint32)])
    assert qdt.itemsize == 24 and rdt.itemsize == 12
    nq = 4000
    q = np.zeros(nq, qdt)
    q["src"] = rng.integers(0, feat_off[-1], nq)
    q["cand"] = rng.integers(0, n_img, nq)
    q["px"], q["py"] = rng.uniform(-100, 1100, nq), rng.uniform(-100, 850, nq)
    q[0] = (int(np.nonzero(perms[0] == 10)[0][0]), 0, feats[0][10][0] + 1, feats[0][10][1] + 1)   # probe = descriptor 10: ties with 11
    q[1] = (0, 0, feats[0][4][0], feats[0][4][1])
    q[2] = (0, 1, feats[1][0][0] + 10, feats[1][0][1])                                             # a single candidate
    res = np.zeros(nq, rdt)
    L.ochip_dense_match.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_void_p]
    assert L.ochip_dense_match(ix, q.ctypes.data, nq, 150.0, res.ctypes.data) == 0
    alld = np.concatenate(sdesc)
    pop = lambda a: np.unpackbits(a.view(np.uint8), axis=-1).sum(-1)
    for i in range(nq):
        ci = int(q["cand"][i])
        d2 = (sloc[ci][:, 0] - q["px"][i]) ** 2 + (sloc[ci][:, 1] - q["py"][i]) ** 2
        inside = np.nonzero(d2 < 150.0 ** 2)[0]
        assert res["nearby"][i] == len(inside), i
        if len(inside) == 0:
            continue
        ham = pop(sdesc[ci][inside] ^ alld[q["src"][i]])
        order = np.sort(ham)
        assert res["bc"][i] == order[0], i
        assert res["sc"][i] == (order[1] if len(order) > 1 else 0xFFFF), i
        if len(order) == 1 or order[0] < order[1]:
            assert res["best"][i] == inside[np.argmin(ham)], i
    assert res["bc"][0] == 0 and res["sc"][0] == 0                      # the tie: second best == best
    d_edge = np.nonzero(perms[0] == 5)[0][0]
    d2 = (sloc[0][d_edge][0] - q["px"][1]) ** 2 + (sloc[0][d_edge][1] - q["py"][1]) ** 2
    assert d2 == 150.0 ** 2                                             # on the circle: not in the disc
    assert res["nearby"][2] == 1 and res["sc"][2] == 0xFFFF
    # a radius that does not fit the index's cells is refused
    assert L.ochip_dense_match(ix, q.ctypes.data, nq, 151.0, res.ctypes.data) != 0
    L.ochip_dense_index_destroy.argtypes = [C.c_void_p]
    L.ochip_dense_index_destroy(ix)
    ctx.close()


def test_dense_link_against_numpy():
    """ochip_dense_link alone, through the C ABI (nearest cameras, projection, disc search, accept rule, union-find), against
    a numpy restatement on random features: the accepted match of every (feature, candidate) slot, the roots, the counts."""
    import ctypes as C
    rng = np.random.default_rng(7)
    L = capi.load()
    ctx = capi.Context(0)
    n_img, cell, W, H, F = 4, 151.0, 1000.0, 750.0, 600.0
    cam_xy = np.array([[0.0, 0.0], [40.0, 5.0], [15.0, 60.0], [70.0, 55.0]])
    cam_pos = np.concatenate([cam_xy, np.full((n_img, 1), 100.0)], 1)
    q_down = np.array([1.0, 0.0, 0.0, 0.0])                               # rotation by pi about x: the camera looks down
    model10 = np.array([F, W / 2, H / 2, 0, 0, 0, 0, 0, W, H])
    feats, descs = [], []
    ground = rng.uniform([-60, -50], [130, 110], (500, 2))                # ground points seen by several cameras
    gdesc = rng.integers(0, 1 << 62, (len(ground), 8), dtype=np.uint64)
    gdesc[:, 7] &= np.uint64((1 << 38) - 1)
    for i in range(n_img):
        d = ground - cam_xy[i]
        px = np.stack([W / 2 + F * d[:, 0] / 100.0, H / 2 - F * d[:, 1] / 100.0], 1)
        ok = (px[:, 0] >= 0) & (px[:, 0] < W) & (px[:, 1] >= 0) & (px[:, 1] < H)
        noise = gdesc[ok].copy()
        flip = rng.integers(0, 448, (ok.sum(), 6))                        # a few flipped bits per view
        for r in range(len(noise)):
            for b in flip[r]:
                noise[r, b // 64] ^= np.uint64(1) << np.uint64(b % 64)
        feats.append(px[ok] + rng.normal(0, 0.5, (ok.sum(), 2)))
        descs.append(noise)
    feat_off, cell_off, grid2, origin2, cell_start, sdesc, sloc, perms = [0], [0], [], [], [], [], [], []
    for loc, d in zip(feats, descs):
        ox, oy = np.floor(loc.min(0))
        ncx, ncy = int((loc[:, 0].max() - ox) // cell) + 1, int((loc[:, 1].max() - oy) // cell) + 1
        c = (np.floor((loc[:, 1] - oy) / cell) * ncx + np.floor((loc[:, 0] - ox) / cell)).astype(int)
        perm = np.argsort(c, kind="stable")
        starts = np.searchsorted(c[perm], np.arange(ncx * ncy + 1))
        perms.append(perm), sdesc.append(d[perm]), sloc.append(loc[perm]), cell_start.append(starts.astype(np.uint32))
        grid2 += [ncx, ncy]
        origin2 += [ox, oy]
        feat_off.append(feat_off[-1] + len(loc))
        cell_off.append(cell_off[-1] + ncx * ncy + 1)
    total = feat_off[-1]
    ix = C.c_void_p()
    u64 = lambda a: np.ascontiguousarray(a, np.uint64)
    L.ochip_dense_index_create.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 7 + [C.c_double, C.POINTER(C.c_void_p)]
    keep = [u64(feat_off), np.ascontiguousarray(np.concatenate(sdesc)), np.ascontiguousarray(np.concatenate(sloc)), u64(cell_off),
            np.ascontiguousarray(np.concatenate(cell_start)), np.array(grid2, np.int32), np.array(origin2, np.float64)]
    assert L.ochip_dense_index_create(ctx.h, n_img, *[a.ctypes.data for a in keep], cell, C.byref(ix)) == 0
    # measurement ids: image offset + feature number in the image's ORIGINAL order
    id_of_pos = np.concatenate([feat_off[i] + perms[i] for i in range(n_img)]).astype(np.uint32)
    # hit points: the ground point under each feature's pixel (z = 0), a few features without one
    hits = np.full((total, 3), np.nan)
    for i in range(n_img):
        loc = sloc[i]
        gx = cam_xy[i][0] + (loc[:, 0] - W / 2) * 100.0 / F
        gy = cam_xy[i][1] - (loc[:, 1] - H / 2) * 100.0 / F
        hits[feat_off[i]:feat_off[i + 1]] = np.stack([gx, gy, np.zeros(len(loc))], 1)
    hits[rng.integers(0, total, 25), 0] = np.nan
    cams17 = np.zeros((n_img, 17))
    cams17[:, 0:3], cams17[:, 3:7], cams17[:, 7:17] = cam_pos, [-q_down[0], -q_down[1], -q_down[2], q_down[3]], model10
    root = np.zeros(total, np.uint32)
    counts = np.zeros(2, np.uint64)
    slot_dst = np.zeros((total, 11), np.uint32)
    L.ochip_dense_link.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint32, C.c_uint32, C.c_double, C.c_double,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
    hits_c = np.ascontiguousarray(hits)
    assert L.ochip_dense_link(ix, cams17.ctypes.data, id_of_pos.ctypes.data, hits_c.ctypes.data, 150.0, 10, 486, 0.85, 0.35, root.ctypes.data,
                              counts.ctypes.data, slot_dst.ctypes.data) == 0, L.ochip_last_error(ctx.h)
    # ---- numpy restatement
    pop = lambda a: np.unpackbits(np.ascontiguousarray(a).view(np.uint8), axis=-1).sum(-1)
    alld, NONE = np.concatenate(sdesc), 0xFFFFFFFF
    parent = np.arange(total)

    def find(x):
        while parent[x] != x:
            x = parent[x]
        return x

    exp_slot = np.full((total, 11), NONE, np.uint32)
    n_queries = n_matches = 0
    matched = np.zeros(total, bool)
    for src in range(n_img):
        for k in range(feat_off[src], feat_off[src + 1]):
            if np.isnan(hits[k, 0]):
                continue
            d2 = ((cam_pos - hits[k]) ** 2).sum(1)
            order = sorted(range(n_img), key=lambda c: (d2[c], c))
            used = 0
            for c in order:
                if c == src:
                    continue
                px = np.array([W / 2 + F * (hits[k, 0] - cam_xy[c][0]) / 100.0, H / 2 - F * (hits[k, 1] - cam_xy[c][1]) / 100.0])
                if not (0 <= px[0] < W and 0 <= px[1] < H):
                    continue
                n_queries += 1
                dd = (sloc[c][:, 0] - px[0]) ** 2 + (sloc[c][:, 1] - px[1]) ** 2
                inside = np.nonzero(dd < 150.0 ** 2)[0]
                if len(inside):
                    ham = pop(sdesc[c][inside] ^ alld[k])
                    o = np.sort(ham)
                    best, second = o[0] / 486.0, (o[1] / 486.0 if len(o) > 1 else np.inf)
                    good = best < 0.85 * second if len(inside) >= 2 else best < 0.35
                    if good:
                        a, b = int(id_of_pos[k]), int(id_of_pos[feat_off[c] + inside[np.argmin(ham)]])
                        exp_slot[k, used] = b
                        n_matches += 1
                        matched[[a, b]] = True
                        ra, rb = find(a), find(b)
                        if ra != rb:
                            parent[max(ra, rb)] = min(ra, rb)
                used += 1
    exp_root = np.array([find(i) if matched[i] else NONE for i in range(total)], np.uint32)
    assert n_matches > 200 and int(counts[0]) == n_queries and int(counts[1]) == n_matches
    assert np.array_equal(slot_dst, exp_slot)
    assert np.array_equal(root, exp_root)
    # arguments the kernels were not built for are refused
    assert L.ochip_dense_link(ix, cams17.ctypes.data, id_of_pos.ctypes.data, hits_c.ctypes.data, 150.0, 9, 486, 0.85, 0.35, root.ctypes.data,
                              counts.ctypes.data, None) != 0
    L.ochip_dense_index_destroy.argtypes = [C.c_void_p]
    L.ochip_dense_index_destroy(ix)
    ctx.close()
