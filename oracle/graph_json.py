"""TEST INFRASTRUCTURE (oracle): the reference's graph.json format restated in plain Python.

Follows /root/reference/src/io/serialize_MeasurementGraph.cpp:204-591 (writer: member order, nodes / edges sorted by
id, a node's edge list sorted, descriptor = base64 of 61 bytes with bit j at byte j >> 3 bit j & 7) and
/root/reference/src/io/deserialize_MeasurementGraph.cpp:30-272 (reader: members looked up by name, `num_sparse_features`
optional, one shared camera model per id).  The text layout is rapidjson's PrettyWriter with kFormatSingleLineArray and
kWriteNanAndInfFlag [3P, not in the reference tree]: 4-space indent, arrays on one line with ", ", doubles through
Grisu2 + Prettify (shortest digits; "1.0", "0.001", "1e-7", "1.5e21"), NaN / Infinity / -Infinity spelled out.

Pinned by the one literal the reference's tests hold for this format: the empty graph's text
(/root/reference/test/test_serialize_deserialize.cpp:13-22) - tests/test_graph_io.py checks it.

A graph here is a plain dict:
  {"nodes": {id: {path, position[3], orientation[4] (x y z w), thumbnail (base64 text), model {id, dimensions[2],
                  focal_length, principal[2], radial_distortion[3], tangential_distortion[2], projection},
                  edges [ids], metadata {...}, features [{location[2], strength, descriptor (8 u64 words)}],
                  num_sparse_features}},
   "edges": {id: {source, dest, matches [[i1, i2, distance]], inlier_matches [[[x, y], [x, y], i1, i2, match_index]],
                  relation[9], relation_type, relative_pose [{score, orientation[4], position[3]}]}}}
"""
import base64
import json
import math
from decimal import Decimal

import numpy as np

DESCRIPTOR_BYTES = (486 + 7) >> 3


def format_double(d):
    """rapidjson Writer::WriteDouble (dtoa.h: Grisu2 digits, then Prettify)."""
    d = float(d)
    if math.isnan(d):
        return "NaN"
    if math.isinf(d):
        return "-Infinity" if d < 0 else "Infinity"
    if d == 0:
        return "-0.0" if math.copysign(1.0, d) < 0 else "0.0"
    sign = "-" if d < 0 else ""
    t = Decimal(repr(abs(d))).as_tuple()   # repr: the shortest digits that read back to d
    digits = "".join(str(x) for x in t.digits).lstrip("0")
    k = t.exponent + (len(digits) - len(digits.rstrip("0")))
    digits = digits.rstrip("0")
    length = len(digits)
    kk = length + k                          # position of the decimal point
    if 0 <= k and kk <= 21:
        return sign + digits + "0" * k + ".0"
    if 0 < kk <= 21:
        return sign + digits[:kk] + "." + digits[kk:]
    if -6 < kk <= 0:
        return sign + "0." + "0" * (-kk) + digits
    if length == 1:
        return sign + digits + "e" + str(kk - 1)
    return sign + digits[0] + "." + digits[1:] + "e" + str(kk - 1)


def format_string(s):
    out = ['"']
    for ch in s:
        o = ord(ch)
        if ch == '"':
            out.append('\\"')
        elif ch == "\\":
            out.append("\\\\")
        elif ch == "\b":
            out.append("\\b")
        elif ch == "\f":
            out.append("\\f")
        elif ch == "\n":
            out.append("\\n")
        elif ch == "\r":
            out.append("\\r")
        elif ch == "\t":
            out.append("\\t")
        elif o < 0x20:
            out.append("\\u%04X" % o)
        else:
            out.append(ch)
    out.append('"')
    return "".join(out)


class Int(int):
    """An integer member (Writer::Int64 / Uint64) as opposed to a double."""


def _emit(value, depth, out):
    """PrettyWriter: `depth` = number of open containers around `value`."""
    if isinstance(value, dict):
        if not value:
            out.append("{}")
            return
        out.append("{")
        first = True
        for k, v in value.items():
            out.append("\n" if first else ",\n")
            first = False
            out.append("    " * (depth + 1) + format_string(str(k)) + ": ")
            _emit(v, depth + 1, out)
        out.append("\n" + "    " * depth + "}")
    elif isinstance(value, (list, tuple)):
        out.append("[")
        for i, v in enumerate(value):
            if i:
                out.append(", ")
            _emit(v, depth + 1, out)
        out.append("]")
    elif isinstance(value, str):
        out.append(format_string(value))
    elif isinstance(value, bool):
        out.append("true" if value else "false")
    elif isinstance(value, (Int, np.integer)) or (isinstance(value, int) and not isinstance(value, bool)):
        out.append(str(int(value)))
    else:
        out.append(format_double(value))


def descriptor_to_base64(words):
    """bitset_to_bytes + Base64encode (serialize_MeasurementGraph.cpp:20-27,442-447)."""
    raw = np.asarray(words, np.uint64).astype("<u8").tobytes()[:DESCRIPTOR_BYTES]
    raw = raw[:60] + bytes([raw[60] & 0x3F])
    return base64.b64encode(raw).decode("ascii")


def descriptor_from_base64(text):
    """Base64decode + bitset_from_bytes (deserialize_MeasurementGraph.cpp:16-24,176-181): 8 u64 words, bits >= 486 zero."""
    raw = base64.b64decode(text)
    assert len(raw) == DESCRIPTOR_BYTES, len(raw)
    raw = raw[:60] + bytes([raw[60] & 0x3F]) + b"\0\0\0"
    return np.frombuffer(raw, "<u8").copy()


def default_metadata():
    nan = float("nan")
    return {"camera_info": {"dimensions": [Int(0), Int(0)], "focal_length_px": nan, "principal": [nan, nan], "make": "",
                            "model": "", "serial_no": "", "lens_make": "", "lens_model": ""},
            "capture_info": {"latitude": nan, "longitude": nan, "altitude": nan, "relative_altitude": nan, "roll": nan,
                             "pitch": nan, "yaw": nan, "accuracy_xy": nan, "accuracy_z": nan, "datum": "", "timestamp": "",
                             "datestamp": ""}}


def write_graph(graph):
    """Serializer<MeasurementGraph>::to_json: the text the reference writes for `graph`."""
    doc = {"version": Int(1), "nodes": {}, "edges": {}}
    for nid in sorted(graph["nodes"]):
        n = graph["nodes"][nid]
        m = n["model"]
        doc["nodes"][str(nid)] = {
            "path": n["path"],
            "position": [float(x) for x in n["position"]],
            "orientation": [float(x) for x in n["orientation"]],
            "thumbnail": n.get("thumbnail", ""),
            "model": {"id": Int(m["id"]), "dimensions": [Int(m["dimensions"][0]), Int(m["dimensions"][1])],
                      "focal_length": float(m["focal_length"]), "principal": [float(x) for x in m["principal"]],
                      "radial_distortion": [float(x) for x in m["radial_distortion"]],
                      "tangential_distortion": [float(x) for x in m["tangential_distortion"]],
                      "projection": m.get("projection", "planar")},
            "edges": [str(e) for e in sorted(n["edges"])],
            "metadata": n.get("metadata") or default_metadata(),
            "features": [{"location": [float(f["location"][0]), float(f["location"][1])], "strength": float(f["strength"]),
                          "descriptor": descriptor_to_base64(f["descriptor"])} for f in n["features"]],
            "num_sparse_features": Int(n["num_sparse_features"]),
        }
    for eid in sorted(graph["edges"]):
        e = graph["edges"][eid]
        doc["edges"][str(eid)] = {
            "source": str(e["source"]), "dest": str(e["dest"]),
            "matches": [[Int(m[0]), Int(m[1]), float(m[2])] for m in e["matches"]],
            "inlier_matches": [[[float(m[0][0]), float(m[0][1])], [float(m[1][0]), float(m[1][1])], Int(m[2]), Int(m[3]), Int(m[4])]
                               for m in e["inlier_matches"]],
            "relation": [float(x) for x in e["relation"]],
            "relation_type": e["relation_type"],
            "relative_pose": [{"score": Int(p["score"]), "orientation": [float(x) for x in p["orientation"]],
                               "position": [float(x) for x in p["position"]]} for p in e["relative_pose"]],
        }
    out = []
    _emit(doc, 0, out)
    return "".join(out)


def read_graph(text):
    """Deserializer<MeasurementGraph>::from_json.  Returns the dict layout above, or None when the document is not a
    version-1 graph.  Node and edge order = the file's order (what the reference's insertion-ordered maps end up with)."""
    try:
        doc = json.loads(text)   # accepts NaN / Infinity / -Infinity like kParseNanAndInfFlag; correctly rounded doubles
    except ValueError:
        return None
    if not isinstance(doc, dict) or doc.get("version") != 1 or isinstance(doc.get("version"), float):
        return None
    graph = {"nodes": {}, "edges": {}}
    models = {}
    for key, n in doc["nodes"].items():
        m = n["model"]
        if m["id"] not in models:   # later copies of a known id are ignored (:86-110)
            models[m["id"]] = {"id": m["id"], "dimensions": list(m["dimensions"]), "focal_length": float(m["focal_length"]),
                               "principal": [float(x) for x in m["principal"]],
                               "radial_distortion": [float(x) for x in m["radial_distortion"]],
                               "tangential_distortion": [float(x) for x in m["tangential_distortion"]],
                               "projection": "planar" if m["projection"] == "planar" else "UNKNOWN"}
        feats = [{"location": [float(f["location"][0]), float(f["location"][1])],
                  "strength": float(np.float32(f["strength"])), "descriptor": descriptor_from_base64(f["descriptor"])}
                 for f in n["features"]]
        graph["nodes"][int(key)] = {
            "path": n["path"], "position": [float(x) for x in n["position"]],
            "orientation": [float(x) for x in n["orientation"]], "thumbnail": n["thumbnail"], "model": models[m["id"]],
            "edges": sorted({int(e) for e in n["edges"]}), "metadata": n["metadata"], "features": feats,
            "num_sparse_features": n.get("num_sparse_features", len(feats)),
        }
    for key, e in doc["edges"].items():
        rt = e["relation_type"]
        poses = [{"score": p["score"], "orientation": [float(x) for x in p["orientation"]],
                  "position": [float(x) for x in p["position"]]} for p in e["relative_pose"]]
        graph["edges"][int(key)] = {
            "source": int(e["source"]), "dest": int(e["dest"]),
            "matches": [[int(m[0]), int(m[1]), float(m[2])] for m in e["matches"]],
            "inlier_matches": [[[float(m[0][0]), float(m[0][1])], [float(m[1][0]), float(m[1][1])], int(m[2]), int(m[3]), int(m[4])]
                               for m in e["inlier_matches"]],
            "relation": [float(x) for x in e["relation"]],
            "relation_type": rt if rt in ("homography", "fundamental_matrix") else "UNKNOWN",
            "relative_pose": poses,
        }
    return graph
