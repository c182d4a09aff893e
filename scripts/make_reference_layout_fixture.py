"""tests/golden/reference_layout_graph.json: a small graph.json TYPED from the reference's writer, field by field
(/root/reference/src/io/serialize_MeasurementGraph.cpp:210-591: member order, string ids, "NaN", arrays on one line, objects
inside arrays, rapidjson::PrettyWriter's indentation with kFormatSingleLineArray).  No serializer is involved - not the
product's (csrc/host/graph_io.cpp), not the oracle's (oracle/graph_json.py): the text below IS the fixture, assembled from the
templates as they stand.  tests/test_graph_io.py reads it with the product and expects the values typed here."""
import os

D0 = "AQAA" + "AAAA" * 19 + "IA=="      # 61 bytes, bits 0 and 485 set (byte j >> 3, bit j & 7; :20-27)
D1 = "AgEA" + "AAAA" * 19 + "AA=="      # bits 1 and 8
D2 = "gAAA" + "AAAA" * 19 + "AA=="      # bit 7
META = '''            "metadata": {
                "camera_info": {
                    "dimensions": [4000, 3000],
                    "focal_length_px": 3000.5,
                    "principal": [2000.0, 1500.0],
                    "make": "ACME",
                    "model": "Mapper 2",
                    "serial_no": "SN-0042",
                    "lens_make": "",
                    "lens_model": "24mm f/2.8"
                },
                "capture_info": {
                    "latitude": 47.123456789,
                    "longitude": 8.5,
                    "altitude": 512.25,
                    "relative_altitude": 100.0,
                    "roll": 0.0,
                    "pitch": -90.0,
                    "yaw": 12.5,
                    "accuracy_xy": NaN,
                    "accuracy_z": NaN,
                    "datum": "WGS-84",
                    "timestamp": "10:11:12",
                    "datestamp": "2024:05:06"
                }
            },
'''


def node(nid, path, pos, ori, edges, feats, sparse):
    f = ", ".join('{\n                    "location": [%s],\n                    "strength": %s,\n'
                  '                    "descriptor": "%s"\n                }' % x for x in feats)
    return ('        "%s": {\n            "path": "%s",\n            "position": [%s],\n            "orientation": [%s],\n'
            '            "thumbnail": "iVBORw0KGgo=",\n            "model": {\n                "id": 7,\n'
            '                "dimensions": [4000, 3000],\n                "focal_length": 3000.5,\n'
            '                "principal": [2000.0, 1500.0],\n                "radial_distortion": [0.02, -0.07, 0.1],\n'
            '                "tangential_distortion": [0.0001, -0.0002],\n                "projection": "planar"\n            },\n'
            '            "edges": [%s],\n' % (nid, path, pos, ori, edges) + META +
            '            "features": [%s],\n            "num_sparse_features": %d\n        }' % (f, sparse))


def edge(eid, s, d, matches, inliers, rel, rtype, poses):
    p = ", ".join('{\n                    "score": %d,\n                    "orientation": [%s],\n'
                  '                    "position": [%s]\n                }' % x for x in poses)
    return ('        "%s": {\n            "source": "%s",\n            "dest": "%s",\n            "matches": [%s],\n'
            '            "inlier_matches": [%s],\n            "relation": [%s],\n            "relation_type": "%s",\n'
            '            "relative_pose": [%s]\n        }' % (eid, s, d, matches, inliers, rel, rtype, p))


NAN_POSE = (0, "NaN, NaN, NaN, NaN", "NaN, NaN, NaN")
NODES = ",\n".join([
    node("11", "/data/IMG_0001.JPG", "10.5, -20.25, 100.0", "1.0, 0.0, 0.0, 0.0", '"5", "6"',
         [("1.5, 2.5", "0.75", D0), ("3999.875, 0.30000000000000004", "0.25", D1)], 1),
    node("22", "/data/IMG_0002.JPG", "35.0, -20.0, 101.5", "NaN, NaN, NaN, NaN", '"5"',
         [("100.0, 200.0", "1.0", D1), ("1e-7, 1.5e-9", "0.5", D2)], 2),
    node("33", '/data/IMG \\"3\\".JPG', "10.0, 25.0, 99.0", "0.0, 0.7071067811865476, 0.0, 0.7071067811865476", '"6"',
         [("7.0, 8.0", "0.125", D2)], 1)])
EDGES = ",\n".join([
    edge("5", "11", "22", "[0, 1, 0.0411522633744856], [1, 0, 0.5]", "[[1.5, 2.5], [1e-7, 1.5e-9], 0, 1, 0]",
         "1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.001, -0.002, 1.0", "homography",
         [(8, "0.0, 0.0, 0.0, 1.0", "1.0, 0.0, 0.0"), (3, "0.1, 0.2, 0.3, 0.9", "0.0, -1.0, 0.5"), NAN_POSE, NAN_POSE]),
    edge("6", "33", "11", "", "", "NaN, NaN, NaN, NaN, NaN, NaN, NaN, NaN, NaN", "UNKNOWN", [NAN_POSE] * 4)])
TEXT = '{\n    "version": 1,\n    "nodes": {\n' + NODES + '\n    },\n    "edges": {\n' + EDGES + '\n    }\n}'

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_layout_graph.json")
    with open(out, "w") as fh:
        fh.write(TEXT)
    print(out, len(TEXT))
