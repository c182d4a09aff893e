#!/usr/bin/env python3
"""Pin of SURVEY.md row a3: compare ochip_akaze_batch with cv::AKAZE itself on the three rendered golden views.

No OpenCV exists in the build image, so the device AKAZE is only checked against this project's own
CPU restatement.  This script closes that gap the day an OpenCV build is at hand anywhere:

  1. here:      python scripts/akaze_pin.py --write-views views/       # views/view_{1,2,3}.pgm, 8-bit grey 640x480
  2. elsewhere: run the reference's own call (src/extract/extract_features.cpp:31-36) on each view and save the result:

        import cv2, numpy as np
        out = {}
        for i in (1, 2, 3):
            img = cv2.imread(f"views/view_{i}.pgm", cv2.IMREAD_GRAYSCALE)
            a = cv2.AKAZE_create(cv2.AKAZE_DESCRIPTOR_MLDB, 486, 3, 5e-5)
            kps, desc = a.detectAndCompute(img, None)
            out[f"kp_{i}"] = np.array([[k.pt[0], k.pt[1], k.size, k.angle, k.response, k.octave, k.class_id] for k in kps])
            out[f"desc_{i}"] = desc                     # (n, 61) uint8
        np.savez("akaze_opencv.npz", **out)

  3. here (GPU): python scripts/akaze_pin.py --compare akaze_opencv.npz

The comparison reports, per view: keypoint counts, how many OpenCV keypoints have a device keypoint at the same evolution
level within 0.01 px (and the worst position / size / angle / response difference among those), and the Hamming distance
of the paired descriptors (0 = bit-identical), and attributes what differs to the classes listed at the top of
oracle/akaze.cpp: `suppression` (a different member of one blob kept - since round 6 the restatement runs OpenCV 4.x's passes,
so this class points at a 3.x build or at a mis-recalled detail), `angle`, `descriptor` (summation order, D4).
Exit code 0 only when every view is identical (positions bit-equal, descriptors equal)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SEEDS = (1, 2, 3)
W, H = 640, 480


def views():
    from opencalibration_amd import synth

    return [synth.render_blobs(W, H, s)[:, :, 0].copy() for s in SEEDS]


def write_views(folder):
    os.makedirs(folder, exist_ok=True)
    for s, img in zip(SEEDS, views()):
        with open(os.path.join(folder, f"view_{s}.pgm"), "wb") as fh:
            fh.write(b"P5\n%d %d\n255\n" % (W, H))
            fh.write(np.ascontiguousarray(img, np.uint8).tobytes())
    print(f"wrote {len(SEEDS)} views to {folder}")


def compare(path):
    from opencalibration_amd import capi

    ref = np.load(path)
    ctx = capi.Context(0)
    imgs = np.stack([np.repeat(v[:, :, None], 3, axis=2) for v in views()])
    got, _ = ctx.akaze_batch(imgs, max_kp=20000)
    identical = True
    for i, s in enumerate(SEEDS):
        gkp, gdesc = got[i]                             # kp6: x, y, diameter, angle (rad), response, evolution level
        rkp, rdesc = np.asarray(ref[f"kp_{s}"], np.float64), np.asarray(ref[f"desc_{s}"], np.uint8)
        gbytes = np.ascontiguousarray(gdesc).view(np.uint8).reshape(len(gdesc), 64)[:, :61]
        paired, worst = 0, np.zeros(4)
        hams = []
        for k in range(len(rkp)):
            same_level = np.flatnonzero(gkp[:, 5] == rkp[k, 6])
            if len(same_level) == 0:
                continue
            d = np.hypot(gkp[same_level, 0] - rkp[k, 0], gkp[same_level, 1] - rkp[k, 1])
            j = same_level[int(np.argmin(d))]
            if d.min() > 0.01:
                continue
            paired += 1
            ang = np.deg2rad(rkp[k, 3])
            worst = np.maximum(worst, [d.min(), abs(gkp[j, 2] - rkp[k, 2]), abs((gkp[j, 3] - ang + np.pi) % (2 * np.pi) - np.pi),
                                       abs(gkp[j, 4] - rkp[k, 4])])
            hams.append(int(np.unpackbits(gbytes[j] ^ rdesc[k]).sum()))
        hams = np.array(hams if hams else [0])
        # ---- attribution to the known departures listed at the top of oracle/akaze.cpp
        #  suppression:      a keypoint of one side without a partner that has a keypoint of the OTHER side at its own or an
        #                    adjacent level within its size - the two sides kept different members of one blob (D1 is removed:
        #                    expected 0 against an OpenCV 4.x build)
        #  angle (D2):       paired, same position, orientation differs by less than the polynomial's 0.3 degrees
        #  descriptor (D2/D4): paired, same position and angle class, descriptor bits differ
        def near_other(kp_a, level_a, kp_b, level_b):
            n = 0
            for k in range(len(kp_a)):
                m = np.abs(level_b - level_a[k]) <= 1
                if np.any(np.hypot(kp_b[m, 0] - kp_a[k, 0], kp_b[m, 1] - kp_a[k, 1]) <= 0.5 * kp_a[k, 2]):
                    n += 1
            return n
        r_un = np.array([k for k in range(len(rkp)) if not np.any((gkp[:, 5] == rkp[k, 6]) & (np.hypot(gkp[:, 0] - rkp[k, 0], gkp[:, 1] - rkp[k, 1]) <= 0.01))], int)
        g_un = np.array([k for k in range(len(gkp)) if not np.any((rkp[:, 6] == gkp[k, 5]) & (np.hypot(rkp[:, 0] - gkp[k, 0], rkp[:, 1] - gkp[k, 1]) <= 0.01))], int)
        d1 = near_other(rkp[r_un][:, :3], rkp[r_un][:, 6], gkp[:, :3], gkp[:, 5]) + near_other(gkp[g_un][:, :3], gkp[g_un][:, 5], rkp[:, :3], rkp[:, 6])
        print(f"   unpaired: OpenCV {len(r_un)}, device {len(g_un)}; {d1} of them are another member of a blob the other side kept (class `suppression`: a keypoint of the other side at an "
              f"adjacent level within their size); angle differences up to {np.rad2deg(worst[2]):.3f} degrees "
              f"({'within' if np.rad2deg(worst[2]) <= 0.3 else 'BEYOND'} the polynomial's 0.3); {int((hams > 0).sum())} paired descriptors differ (D2 / D4)")
        same = paired == len(rkp) == len(gkp) and worst.max() == 0 and hams.max() == 0
        identical &= same
        print(f"view {s}: OpenCV {len(rkp)} keypoints, device {len(gkp)}, paired {paired}; worst |dx| {worst[0]:.3g} px, "
              f"|dsize| {worst[1]:.3g}, |dangle| {worst[2]:.3g} rad, |dresponse| {worst[3]:.3g}; descriptor Hamming "
              f"median {np.median(hams):.0f}, max {hams.max()} of 486 bits, {int((hams == 0).sum())} identical"
              f"{'  == IDENTICAL' if same else ''}")
    ctx.close()
    return 0 if identical else 1


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--write-views", metavar="DIR")
    ap.add_argument("--compare", metavar="NPZ")
    args = ap.parse_args()
    if args.write_views:
        write_views(args.write_views)
    if args.compare:
        sys.exit(compare(args.compare))
    if not args.write_views and not args.compare:
        ap.print_help()


if __name__ == "__main__":
    main()
