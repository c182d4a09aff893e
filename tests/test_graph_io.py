"""graph.json / surface PLY / checkpoint directory of the host library against the plain-Python
restatement of the reference's format (oracle/graph_json.py), the one literal the reference's tests hold for it (the empty
graph's text, test/test_serialize_deserialize.cpp:13-22) and the reference's checkpoint test cases
(test/test_checkpoint.cpp:28-235) restated.  No device involved."""
import json
import os

import numpy as np
import pytest

from opencalibration_amd import host
from oracle import graph_json as gj

EMPTY_GRAPH_TEXT = "{\n    \"version\": 1,\n    \"nodes\": {},\n    \"edges\": {}\n}"   # the reference's literal

MODEL_A = [3000.0, 2000.0, 1500.0, 0.02, -0.07, 0.1, 1e-4, -2e-4, 4000, 3000]
MODEL_B = [2500.5, 1000.25, 700.125, 0.0, 0.0, 0.0, 0.0, 0.0, 2000, 1500]


def _random_graph(seed=3, n_nodes=4, n_feat=40):
    """A host graph with features, orientations, two camera models and edges carrying every field the format holds."""
    rng = np.random.default_rng(seed)
    g = host.Graph()
    ma, mb = g.add_model(MODEL_A), g.add_model(MODEL_B)
    feats, ids = [], []
    for i in range(n_nodes):
        k = n_feat + 3 * i
        loc = rng.uniform(0, 4000, (k, 2))
        loc[0] = [0.1 + 0.2, 1e-7]               # digits that need 17 significant figures / the exponent form
        st = rng.uniform(1e-5, 1, k).astype(np.float32)
        desc = rng.integers(0, 2 ** 63, (k, 8), dtype=np.uint64) * 2 + rng.integers(0, 2, (k, 8), dtype=np.uint64)
        desc[:, 7] &= np.uint64((1 << 38) - 1)   # 486 = 7 * 64 + 38 bits
        pos = rng.normal(size=3) * [100, 100, 5] + [0, 0, 100]
        ids.append(g.add_image(loc, st, desc, k // 2, ma if i % 2 == 0 else mb, pos))
        feats.append((loc, st, desc, pos))
    ori = rng.normal(size=(n_nodes, 4))
    ori /= np.linalg.norm(ori, axis=1, keepdims=True)
    ori[n_nodes - 1] = np.nan                    # a camera that has no orientation yet
    g.set_orientations(ori)
    edges = []
    for (a, b) in [(0, 1), (1, 2), (0, 3)]:
        if max(a, b) >= n_nodes:
            continue
        m = 12
        f1, f2 = rng.permutation(n_feat)[:m], rng.permutation(n_feat)[:m]
        dist = rng.integers(0, 100, m) / 486.0
        n_in = 7
        px = np.concatenate([feats[a][0][f1[:n_in]], feats[b][0][f2[:n_in]]], axis=1)
        H = rng.normal(size=(3, 3))
        poses = rng.normal(size=(4, 8))
        poses[:, 7] = rng.integers(0, 9, 4)
        poses[3, :7] = np.nan
        eid = g.add_edge(ids[a], ids[b], px, f1[:n_in], f2[:n_in], match_index=np.arange(n_in), H=H, dist=dist, poses=poses,
                         match_idx=np.stack([f1, f2], axis=1))
        edges.append(dict(id=eid, src=ids[a], dst=ids[b], f1=f1, f2=f2, dist=dist, px=px, H=H, poses=poses, n_in=n_in))
    for i in range(n_nodes):
        g.set_node_path(i, "/data/survey/IMG_%04d \"x\".JPG" % i)
    return g, ids, feats, ori, edges


def test_empty_graph_text_is_the_reference_literal():
    assert host.Graph().to_json() == EMPTY_GRAPH_TEXT
    assert gj.write_graph({"nodes": {}, "edges": {}}) == EMPTY_GRAPH_TEXT
    g = host.Graph().from_json(EMPTY_GRAPH_TEXT)
    assert g.num_nodes == 0 and g.num_edges == 0


def test_written_graph_reads_back_with_an_independent_reader():
    g, ids, feats, ori, edges = _random_graph()
    text = g.to_json()
    doc = gj.read_graph(text)
    assert doc is not None and sorted(doc["nodes"]) == sorted(ids) and list(doc["nodes"]) == sorted(ids)
    for i, nid in enumerate(ids):
        n = doc["nodes"][nid]
        loc, st, desc, pos = feats[i]
        assert np.array_equal(np.array([f["location"] for f in n["features"]]), loc)          # bit-exact doubles
        assert np.array_equal(np.array([f["strength"] for f in n["features"]], np.float32), st)
        assert np.array_equal(np.array([f["descriptor"] for f in n["features"]]), desc)
        assert np.array_equal(n["position"], pos)
        assert np.array_equal(n["orientation"], ori[i], equal_nan=True)
        assert n["num_sparse_features"] == len(loc) // 2
        assert n["path"] == "/data/survey/IMG_%04d \"x\".JPG" % i
        m = MODEL_A if i % 2 == 0 else MODEL_B
        assert n["model"]["focal_length"] == m[0] and n["model"]["principal"] == m[1:3]
        assert n["model"]["radial_distortion"] == m[3:6] and n["model"]["tangential_distortion"] == m[6:8]
        assert n["model"]["dimensions"] == [m[8], m[9]] and n["model"]["projection"] == "planar"
        assert n["edges"] == sorted(e["id"] for e in edges if nid in (e["src"], e["dst"]))
    assert list(doc["edges"]) == sorted(e["id"] for e in edges)
    for e in edges:
        d = doc["edges"][e["id"]]
        assert (d["source"], d["dest"]) == (e["src"], e["dst"]) and d["relation_type"] == "homography"
        assert np.array_equal([m[:2] for m in d["matches"]], np.stack([e["f1"], e["f2"]], axis=1))
        assert np.array_equal([m[2] for m in d["matches"]], e["dist"])
        assert np.array_equal([m[0] + m[1] for m in d["inlier_matches"]], e["px"])
        assert np.array_equal([m[2:] for m in d["inlier_matches"]],
                              np.stack([e["f1"][:e["n_in"]], e["f2"][:e["n_in"]], np.arange(e["n_in"])], axis=1))
        assert np.array_equal(d["relation"], e["H"].reshape(9))
        for k, p in enumerate(d["relative_pose"]):
            assert np.array_equal(p["orientation"], e["poses"][k, :4], equal_nan=True)
            assert np.array_equal(p["position"], e["poses"][k, 4:7], equal_nan=True) and p["score"] == int(e["poses"][k, 7])
    # the text itself is what the restated writer produces from the parsed document (layout, member order, number format)
    assert gj.write_graph(doc) == text


def test_serialize_deserialize_serialize_is_stable():
    """test_serialize_deserialize.cpp:25-63: serialized == serialize(deserialize(serialized))."""
    g, ids, *_ = _random_graph(seed=9)
    text = g.to_json()
    g2 = host.Graph().from_json(text)
    assert g2.to_json() == text
    assert g2.node_ids == sorted(ids)            # iteration order becomes the file's (sorted) order
    t1, t2 = g.node_table(), g2.node_table()
    order = np.argsort(t1["id"])
    assert np.array_equal(t1["features"][order], t2["features"]) and np.array_equal(t1["sparse"][order], t2["sparse"])
    assert len(g2.models()) == 2                 # one shared model per id
    for i2, i1 in enumerate(order):
        a, b = g.node_payload(int(i1)), g2.node_payload(i2)
        for k in ("loc", "strength", "desc", "position"):
            assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(a["orientation"], b["orientation"], equal_nan=True) and a["path"] == b["path"]


def test_reads_a_reference_style_file():
    """A document the reference could have written (restated writer): 64-bit ids, metadata, a thumbnail, members in another
    order, a member the reader does not know, and no num_sparse_features (files from before the field existed)."""
    rng = np.random.default_rng(5)
    model = {"id": 7, "dimensions": [4000, 3000], "focal_length": 2997.123456789, "principal": [1999.5, 1501.25],
             "radial_distortion": [0.01, -0.02, 0.003], "tangential_distortion": [1e-5, -1e-5]}
    meta = gj.default_metadata()
    meta["camera_info"].update(dimensions=[gj.Int(4000), gj.Int(3000)], focal_length_px=3000.0, make="DJI", model="FC6310")
    meta["capture_info"].update(latitude=47.3769, longitude=8.5417, altitude=512.25, datum="WGS-84", timestamp="12:00:01")
    nodes = {}
    ids = [int(x) for x in rng.integers(1 << 40, 1 << 63, 3, dtype=np.uint64)]
    eid = int(rng.integers(1 << 40, 1 << 63, dtype=np.uint64))
    for i, nid in enumerate(ids):
        feats = [{"location": [float(rng.uniform(0, 4000)), float(rng.uniform(0, 3000))], "strength": float(np.float32(rng.uniform())),
                  "descriptor": np.concatenate([rng.integers(0, 1 << 63, 7, dtype=np.uint64), [np.uint64(123456789)]])} for _ in range(5)]
        nodes[nid] = dict(path="img%d.jpg" % i, position=[1.5 * i, -2.0, 100.0], orientation=[0.0, 1.0, 0.0, 6.123233995736766e-17],
                          thumbnail="iVBORw0KGgo=", model=model, edges=[eid] if i < 2 else [], metadata=meta, features=feats,
                          num_sparse_features=3)
    edges = {eid: dict(source=ids[0], dest=ids[1], matches=[[0, 1, 0.125], [2, 3, 40 / 486.0]],
                       inlier_matches=[[[1.0, 2.0], [3.0, 4.5], 0, 1, 0]], relation=[1, 0, 0, 0, 1, 0, 0, 0, 1.0],
                       relation_type="homography",
                       relative_pose=[dict(score=3, orientation=[0, 0, 0, 1.0], position=[0.1, 0.2, 0.3])] * 4)}
    text = gj.write_graph({"nodes": nodes, "edges": edges})
    g = host.Graph().from_json(text)
    assert g.to_json() == text                                   # byte-identical re-serialisation, metadata included
    doc = json.loads(text)
    # same content, another member order, an unknown member, no num_sparse_features
    shuffled = {"edges": doc["edges"], "generator": {"name": "x", "v": [1, 2, {"a": None}]}, "nodes": {}, "version": 1}
    for k, n in doc["nodes"].items():
        n = dict(reversed(list(n.items())))
        del n["num_sparse_features"]
        shuffled["nodes"][k] = n
    g2 = host.Graph().from_json(json.dumps(shuffled))
    tab = g2.node_table()
    assert list(tab["id"]) == sorted(ids) and list(tab["features"]) == [5, 5, 5] and list(tab["sparse"]) == [5, 5, 5]
    ref = gj.read_graph(text)
    for i, nid in enumerate(sorted(ids)):
        p = g2.node_payload(i)
        assert np.array_equal(p["desc"], np.array([f["descriptor"] for f in ref["nodes"][nid]["features"]]))
        assert np.array_equal(p["loc"], np.array([f["location"] for f in ref["nodes"][nid]["features"]]))
    m = g2.models()
    assert len(m) == 1 and m[0, 10] == 7 and m[0, 0] == 2997.123456789 and list(m[0, 8:10]) == [4000, 3000]
    e = g2.edges(with_distances=True)[0]
    assert (e["source"], e["dest"]) == (ids[0], ids[1]) and e["is_homography"] and list(e["dist"]) == [0.125, 40 / 486.0]
    assert np.array_equal(e["px"], [[1.0, 2.0, 3.0, 4.5]]) and np.array_equal(e["match_idx"], [[0, 1], [2, 3]])


@pytest.mark.parametrize("text", [
    "", "[]", "{\"version\": 2, \"nodes\": {}, \"edges\": {}}", "{\"version\": 1.0, \"nodes\": {}, \"edges\": {}}",
    "{\"version\": 1, \"nodes\": {}}", "{\"version\": 1, \"nodes\": {}, \"edges\": {}} x",
    "{\"version\": 1, \"nodes\": {\"5\": {\"path\": \"a\"}}, \"edges\": {}}",
    EMPTY_GRAPH_TEXT[:-3],
])
def test_rejects_what_is_not_a_version_1_graph(text):
    g, ids, *_ = _random_graph(n_nodes=2)
    with pytest.raises(ValueError):
        g.from_json(text)
    assert g.num_nodes == 2                        # a failed load leaves the graph alone


def test_non_finite_numbers_round_trip():
    g = host.Graph()
    m = g.add_model(MODEL_B)
    g.add_image(np.array([[np.inf, -np.inf]]), np.array([1.0], np.float32), np.zeros((1, 8), np.uint64), 1, m, [np.nan, 0.0, -0.0])
    text = g.to_json()
    assert "[Infinity, -Infinity]" in text and "\"position\": [NaN, 0.0, -0.0]" in text
    p = host.Graph().from_json(text).node_payload(0)
    assert np.array_equal(p["loc"], [[np.inf, -np.inf]]) and np.isnan(p["position"][0]) and np.signbit(p["position"][2])


# ---- surface PLY ---------------------------------------------------------------------------------------------------
def test_mesh_ply_round_trip(tmp_path):
    """test_serialize_deserialize.cpp:65-127: write, read back equal, write again byte-identical."""
    cams = np.array([[0, 0, 0], [1, 0, 0.5], [1, 1, 0.3], [0, 1, -0.5]], float) * 40
    s = host.rebuild_mesh(cams)
    a = s.arrays()
    assert len(a["vertices"]) >= 4 and len(a["edges"]) >= 5
    p1, p2 = tmp_path / "surface.ply", tmp_path / "surface2.ply"
    s.save_ply(p1)
    s2 = host.Surface().load_ply(p1)
    b = s2.arrays()
    assert np.allclose(a["vertices"], b["vertices"], rtol=1e-5, atol=1e-12)     # `ostream << double`: 6 digits
    border = a["edges"][:, 2] == 1
    assert np.array_equal(a["edges"][:, :4], b["edges"][:, :4])
    assert np.array_equal(a["edges"][~border, 4], b["edges"][~border, 4])
    s2.save_ply(p2)
    s3 = host.Surface().load_ply(p2)
    s3.save_ply(tmp_path / "surface3.ply")
    assert open(p2).read() == open(tmp_path / "surface3.ply").read()
    lines = open(p1).read().split("\n")
    assert lines[:4] == ["ply", "format ascii 1.0", "comment exported from OpenCalibration", "element vertex %d" % len(a["vertices"])]
    n_faces = int([l for l in lines if l.startswith("element face ")][0].split()[-1])
    assert n_faces == (2 * (len(a["edges"]) - border.sum()) + border.sum()) // 3    # every triangle once


def test_reads_a_reference_ply_with_64_bit_ids(tmp_path):
    """The reference's mesh ids are random 64-bit numbers, written in sorted order; corners are node ids."""
    ids = [11400714819323198485, 2 ** 62 + 12345, 2 ** 63 + 99, 977]
    order = sorted(range(4), key=lambda i: ids[i])
    xyz = {ids[0]: (0, 0, 1), ids[1]: (10, 0, 2), ids[2]: (10, 10, 3), ids[3]: (0, 10, 4)}
    seq = {ids[i]: k for k, i in enumerate(order)}
    # two triangles (0,1,2), (0,2,3): 4 border edges + the diagonal
    edges = [(ids[0], ids[1], 1, ids[2], 0), (ids[1], ids[2], 1, ids[0], 0), (ids[2], ids[3], 1, ids[0], 0),
             (ids[3], ids[0], 1, ids[2], 0), (ids[0], ids[2], 0, ids[1], ids[3])]
    text = ["ply", "format ascii 1.0", "comment exported from OpenCalibration", "element vertex 4", "property double x",
            "property double y", "property double z", "property int nodeIndex", "element face 2",
            "property list uchar int vertex_index", "element edge 5", "property int vertex1", "property int vertex2",
            "property int edgeIndex", "property uchar border", "property int oppositeCorner1", "property int oppositeCorner2",
            "end_header"]
    for i in order:
        text.append("%g %g %g %d" % (*xyz[ids[i]], ids[i]))
    text += ["3 0 1 2", "3 0 2 3"]
    for k, (s, d, b, o1, o2) in enumerate(edges):
        text.append("%d %d %d %d %d %d" % (seq[s], seq[d], 1000 + k, b, o1, o2))
    path = tmp_path / "ref.ply"
    path.write_text("\n".join(text) + "\n")
    a = host.Surface().load_ply(path).arrays()
    assert np.array_equal(a["vertices"], [xyz[ids[i]] for i in order])
    for k, (s, d, b, o1, o2) in enumerate(edges):
        assert list(a["edges"][k, :4]) == [seq[s], seq[d], b, seq[o1]]
        if not b:
            assert a["edges"][k, 4] == seq[o2]
    for bad in (text[:-1] + ["0 1 7"], text + ["trailing"], ["plx"] + text[1:]):
        path.write_text("\n".join(bad) + "\n")
        with pytest.raises(IOError):
            host.Surface().load_ply(path)


# ---- checkpoint directory (test/test_checkpoint.cpp restated) -------------------------------------------------------------
def test_checkpoint_save_and_load_empty(tmp_path):
    d = tmp_path / "cp"
    host.save_checkpoint(d, host.Graph(), state="INITIAL_PROCESSING", state_run_count=0, origin=(47.3769, 8.5417))
    assert host.validate_checkpoint(d)
    g, surfaces, meta = host.load_checkpoint(d)
    assert g.num_nodes == 0 and surfaces == [] and meta == dict(state="INITIAL_PROCESSING", state_run_count=0, origin=(47.3769, 8.5417))
    assert json.load(open(d / "metadata.json")) == {"version": 1, "state": "INITIAL_PROCESSING", "state_run_count": 0,
                                                    "origin_latitude": 47.3769, "origin_longitude": 8.5417, "surface_count": 0}
    assert open(d / "graph.json").read() == EMPTY_GRAPH_TEXT


def test_checkpoint_save_and_load_with_surfaces(tmp_path):
    g, ids, *_ = _random_graph(seed=11)
    s0 = host.rebuild_mesh(np.array([[0, 0, 50], [100, 0, 50], [100, 100, 50], [0, 100, 50.0]]))
    s0.set_clouds([[[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]], [[7.0, 8.0, 9.0]]])
    s1 = host.Surface().set_clouds([[[10.0, 11.0, 12.0]]])
    d = tmp_path / "cp"
    host.save_checkpoint(d, g, [s0, s1], state="MESH_REFINEMENT", state_run_count=5, origin=(1.0, 2.0))
    assert sorted(os.listdir(d)) == ["graph.json", "metadata.json", "pointcloud_0_0.xyz", "pointcloud_0_1.xyz", "pointcloud_1_0.xyz",
                                     "surface_0.ply", "surface_0_cloudcount.txt", "surface_1_cloudcount.txt"]
    assert open(d / "pointcloud_0_0.xyz").read() == "1,2,3\n4,5,6\n" and open(d / "surface_0_cloudcount.txt").read() == "2"
    g2, surfaces, meta = host.load_checkpoint(d)
    assert meta == dict(state="MESH_REFINEMENT", state_run_count=5, origin=(1.0, 2.0)) and g2.to_json() == g.to_json()
    assert len(surfaces) == 2
    c0, c1 = surfaces[0].clouds(), surfaces[1].clouds()
    assert [len(c) for c in c0] == [2, 1] and np.array_equal(c0[0][0], [1, 2, 3]) and np.array_equal(c0[1][0], [7, 8, 9])
    assert [len(c) for c in c1] == [1] and np.array_equal(c1[0][0], [10, 11, 12])
    assert len(surfaces[0].arrays()["vertices"]) == len(s0.arrays()["vertices"]) and len(surfaces[1].arrays()["vertices"]) == 0


def test_checkpoint_failures(tmp_path):
    assert not host.validate_checkpoint("/nonexistent/path/to/checkpoint")
    with pytest.raises(IOError):
        host.load_checkpoint("/nonexistent/path/to/checkpoint")
    d = tmp_path / "cp"
    d.mkdir()
    (d / "graph.json").write_text(EMPTY_GRAPH_TEXT)
    assert not host.validate_checkpoint(d)                                    # no metadata.json
    (d / "metadata.json").write_text("{ this is not valid json }")
    with pytest.raises(IOError):
        host.load_checkpoint(d)
    (d / "metadata.json").write_text("{\"version\": 999}")
    with pytest.raises(IOError, match="version"):
        host.load_checkpoint(d)
    (d / "metadata.json").write_text("{\"version\": 1, \"state\": \"NOT_A_STATE\", \"state_run_count\": 2, \"origin_latitude\": 0, "
                                     "\"origin_longitude\": 0, \"surface_count\": 0}")
    assert host.load_checkpoint(d)[2]["state"] == "INITIAL_PROCESSING"        # unknown names fall back (checkpoint.cpp:86-87)
    os.remove(d / "graph.json")
    assert not host.validate_checkpoint(d)
    with pytest.raises(IOError):
        host.load_checkpoint(d)


def test_reads_a_graph_typed_from_the_reference_writer():
    """tests/golden/reference_layout_graph.json was typed from serialize_MeasurementGraph.cpp:210-591 field by field
    (scripts/make_reference_layout_fixture.py holds the templates; no serializer wrote it): three nodes, two edges, metadata,
    thumbnails, NaN orientations, an escaped path, exponent-form doubles.  The product's reader must find every value, and
    its writer must give the text back byte for byte."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_layout_graph.json")
    text = open(path).read()
    g = host.Graph().load_json(path)
    assert g.node_ids == [11, 22, 33] and g.num_edges == 2
    tab = g.node_table()
    assert tab["features"].tolist() == [2, 2, 1] and tab["sparse"].tolist() == [1, 2, 1] and tab["model"].tolist() == [0, 0, 0]
    m = g.models()
    assert len(m) == 1 and m[0].tolist() == [3000.5, 2000.0, 1500.0, 0.02, -0.07, 0.1, 0.0001, -0.0002, 4000, 3000, 7]
    n0, n1, n2 = (g.node_payload(i) for i in range(3))
    assert n0["path"] == "/data/IMG_0001.JPG" and n2["path"] == '/data/IMG "3".JPG'
    assert n0["position"].tolist() == [10.5, -20.25, 100.0] and n0["orientation"].tolist() == [1.0, 0.0, 0.0, 0.0]
    assert np.isnan(n1["orientation"]).all() and n2["orientation"].tolist() == [0.0, 0.7071067811865476, 0.0, 0.7071067811865476]
    assert n0["loc"].tolist() == [[1.5, 2.5], [3999.875, 0.1 + 0.2]] and n0["strength"].tolist() == [0.75, 0.25]
    assert n1["loc"].tolist() == [[100.0, 200.0], [1e-7, 1.5e-9]]
    bits = lambda d: [j for j in range(512) if (int(d[j >> 6]) >> (j & 63)) & 1]
    assert bits(n0["desc"][0]) == [0, 485] and bits(n0["desc"][1]) == [1, 8] and bits(n1["desc"][1]) == [7]
    e0, e1 = g.edges(with_distances=True)
    assert (e0["source"], e0["dest"], e0["n_matches"], e0["n_inliers"], e0["is_homography"]) == (11, 22, 2, 1, True)
    assert e0["match_idx"].tolist() == [[0, 1], [1, 0]] and e0["dist"].tolist() == [20 / 486.0, 0.5]
    assert e0["px"].tolist() == [[1.5, 2.5, 1e-7, 1.5e-9]] and (e0["f1"][0], e0["f2"][0], e0["match_index"][0]) == (0, 1, 0)
    assert e0["H"].tolist() == [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.001, -0.002, 1.0]]
    assert e0["poses"][0].tolist() == [0, 0, 0, 1, 1, 0, 0, 8] and e0["poses"][1].tolist() == [0.1, 0.2, 0.3, 0.9, 0.0, -1.0, 0.5, 3]
    assert np.isnan(e0["poses"][2, :7]).all() and e0["poses"][2, 7] == 0
    assert (e1["source"], e1["dest"], e1["n_matches"], e1["n_inliers"], e1["is_homography"]) == (33, 11, 0, 0, False)
    assert np.isnan(e1["H"]).all()
    assert g.to_json() == text                                   # member order, indentation, number format, metadata, thumbnail
    # Python's own JSON reader agrees on the structure (NaN is accepted by it) and the oracle's reader on the values
    doc = json.loads(text)
    assert list(doc) == ["version", "nodes", "edges"] and list(doc["nodes"]["11"])[:4] == ["path", "position", "orientation", "thumbnail"]
    assert doc["nodes"]["22"]["metadata"]["capture_info"]["datum"] == "WGS-84" and doc["nodes"]["33"]["edges"] == ["6"]
    od = gj.read_graph(text)
    assert od is not None and gj.write_graph(od) == text
