#!/usr/bin/env python3
"""Where the keypoints of a rendered view live: count per pyramid level and per 64 x 24 detection tile (the numbers the
descriptor kernel's design is sized by).  usage: probe_kp_levels.py [blob spacing = 21] [views = 2]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    os.environ["OCHIP_BLOB_SPACING"] = sys.argv[1]
import numpy as np

from opencalibration_amd import capi, pipeline, synth


def main():
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cfg = synth.CONFIGS["C2"]
    grid = synth.make_grid(seed=12345, rows=cfg["rows"], cols=cfg["cols"], feats=64)
    ctx = capi.Context(0)
    images, (cnt, h, w) = pipeline.synthetic_views(ctx, grid, block=(0, n))
    views = np.stack([ctx.synth_views_read(images, i, w, h) for i in range(n)])
    res, (W, H) = ctx.akaze_batch(views, max_kp=60000)
    for i, (kp, _) in enumerate(res):
        lvl = kp[:, 5].astype(int)
        print(f"view {i}: {len(kp)} keypoints, working image {W} x {H}; per level:", np.bincount(lvl, minlength=16).tolist())
        for level in range(4):
            sel = kp[lvl == level]
            tx, ty = (sel[:, 0] // 64).astype(int), (sel[:, 1] // 24).astype(int)
            tiles = np.zeros(((H + 23) // 24, (W + 63) // 64), int)
            np.add.at(tiles, (ty, tx), 1)
            print(f"  level {level}: per 64x24 tile mean {tiles.mean():.2f} max {tiles.max()} empty {np.mean(tiles == 0):.2f}")
    ctx.synth_views_free(images)
    ctx.close()


if __name__ == "__main__":
    main()
