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
                                dict(seed=8, n_points=700, distractors=5, rows=2, cols=3)])
def test_densify_matches_the_restatement(kw):
    scene = dense_scene(**kw)
    ctx = capi.Context(0)
    exp, got, surface, exp_surface, g = _both(scene, ctx)
    assert got["matches"] == exp["matches"] > 300
    assert np.array_equal(got["match_pairs"], exp["match_pairs"])               # same matches, same order
    assert got["tracks"] >= exp["tracks"] and got["points"] == len(exp["points"])
    cloud = surface.clouds()[-1]
    assert np.allclose(cloud, exp["points"], rtol=0, atol=1e-9)
    dz = cloud[:, 2] - scene["ground"](cloud[:, 0], cloud[:, 1])
    assert np.median(np.abs(dz)) < 0.2
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
