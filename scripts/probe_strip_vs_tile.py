#!/usr/bin/env python3
"""Which keypoints of a view get different descriptors from the strip kernels than from the tile kernels
(OCHIP_TEST_HOOKS=tile_levels,tile_det), and where they sit.  usage: probe_strip_vs_tile.py [w=2000] [h=1500]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def run(tag):
    from opencalibration_amd import capi, synth
    w, h = int(sys.argv[2]), int(sys.argv[3])
    img = synth.render_blobs(w, h, 9)
    if os.environ.get("PROBE_NOISE"):
        rng = np.random.default_rng(w)
        base = synth.render_blobs(w, h, 31)
        img = np.clip(base.astype(np.int32) + rng.integers(0, 40, (h, w, 1)) - 20, 0, 255).astype(np.uint8)
    if os.environ.get("PROBE_TINT"):
        rng = np.random.default_rng(5)
        tint = rng.integers(0, 40, (h, w, 3), dtype=np.uint8)
        img = np.clip(img.astype(np.int32) + tint - 20, 0, 255).astype(np.uint8)
    ctx = capi.Context(0)
    os.environ["OCHIP_DUMP_PLANES"] = "/tmp/probe_%s" % tag
    res, wh = ctx.akaze_batch(img[None], max_kp=60000)
    kp, desc = res[0]
    np.savez("/tmp/probe_%s.npz" % tag, kp=kp, desc=desc, wh=np.array(wh))
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("strip", "tile"):
        run(sys.argv[1])
        sys.exit(0)
    w = sys.argv[1] if len(sys.argv) > 1 else "2000"
    h = sys.argv[2] if len(sys.argv) > 2 else "1500"
    for tag, hooks in (("strip", ""), ("tile", "tile_levels,tile_det")):
        env = dict(os.environ, OCHIP_TEST_HOOKS=hooks)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), tag, w, h], env=env)
    a, b = np.load("/tmp/probe_strip.npz"), np.load("/tmp/probe_tile.npz")
    W, H = a["wh"]
    print("working size", W, H, "keypoints", len(a["kp"]), len(b["kp"]), "kp equal", np.array_equal(a["kp"], b["kp"]))
    # the pyramids: where do they differ?
    sizes = [(W >> o, H >> o) for o in range(4) for _ in range(4)]
    for name, comps in (("lt", 1), ("lxy", 2)):
        pa = np.fromfile("/tmp/probe_strip_%s.f32" % name, np.float32)
        pb = np.fromfile("/tmp/probe_tile_%s.f32" % name, np.float32)
        off = 0
        for lvl, (lw, lh) in enumerate(sizes):
            n = lw * lh * comps
            da = pa[off:off + n].reshape(lh, lw, comps).view(np.uint32)
            db = pb[off:off + n].reshape(lh, lw, comps).view(np.uint32)
            off += n
            bad = np.argwhere(np.any(da != db, axis=2))
            if len(bad) and name == "lt" and lvl == 1:
                fa, fb = da.view(np.float32)[:, :, 0], db.view(np.float32)[:, :, 0]
                mid = lh // 2
                cols = np.flatnonzero(da[mid, :, 0] != db[mid, :, 0])
                print("   row %d differing cols %s" % (mid, cols.tolist()))
                for c in cols[:8]:
                    print("     col %d strip %.9g tile %.9g" % (c, fa[mid, c], fb[mid, c]))
                rows = np.flatnonzero(da[:, lw // 2, 0] != db[:, lw // 2, 0])
                print("   col %d differing rows %s" % (lw // 2, rows.tolist()))
                for r_ in rows[:8]:
                    print("     row %d strip %.9g tile %.9g" % (r_, fa[r_, lw // 2], fb[r_, lw // 2]))
            if len(bad):
                print("  %s level %d (%d x %d): %d pixels differ; rows %d..%d cols %d..%d; first %s" %
                      (name, lvl, lw, lh, len(bad), bad[:, 0].min(), bad[:, 0].max(), bad[:, 1].min(), bad[:, 1].max(), bad[:12].tolist()))
    fa = np.fromfile("/tmp/probe_strip_flow1.f32", np.float32).reshape(H, W)
    fb = np.fromfile("/tmp/probe_tile_flow1.f32", np.float32).reshape(H, W)
    badf = np.argwhere(fa.view(np.uint32) != fb.view(np.uint32))
    print("flow level 1: %d pixels differ" % len(badf))
    if len(badf):
        print("   rows %d..%d cols %d..%d" % (badf[:, 0].min(), badf[:, 0].max(), badf[:, 1].min(), badf[:, 1].max()), "distinct cols", np.unique(badf[:, 1]).tolist()[:20], "distinct rows", np.unique(badf[:, 0]).tolist()[:20])
        for r_, c_ in badf[:6]:
            print("     (%d, %d) strip %.9g tile %.9g" % (r_, c_, fa[r_, c_], fb[r_, c_]))
    if len(a["kp"]) == len(b["kp"]):
        bad = np.flatnonzero(np.any(a["desc"] != b["desc"], axis=1))
        print("descriptors differ:", len(bad))
        for i in bad[:40]:
            x, y, size, ang, resp, lvl = a["kp"][i]
            sc = 1 << (int(lvl) // 4)
            bits = sum(bin(int(v)).count("1") for v in (a["desc"][i] ^ b["desc"][i]))
            print("  level %2d  x %7.1f y %7.1f (level px %6.1f %6.1f of %d x %d)  size %5.1f  bits %d" % (lvl, x, y, x / sc, y / sc, W // sc, H // sc, size, bits))
