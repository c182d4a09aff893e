#!/usr/bin/env python3
"""The matcher alone: N images of F random descriptors, every image against its K next neighbours in both directions,
a few launches - the quick workload for traces and counter passes of the match kernels.
usage: match_only.py [N=128] [F=3500] [K=5] [repeats=4]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from opencalibration_amd import capi


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    f = int(sys.argv[2]) if len(sys.argv) > 2 else 3500
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    repeats = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    rng = np.random.default_rng(1)
    ctx = capi.Context(0)
    ctx.descriptors_reserve(n, n * f)
    for i in range(n):
        d = rng.integers(0, 1 << 63, (f, 8), dtype=np.uint64) * 2 + rng.integers(0, 2, (f, 8), dtype=np.uint64)
        d[:, 7] &= np.uint64((1 << (486 - 448)) - 1)      # bits 486..511 are zero in a 486-bit descriptor
        ctx.upload_descriptors(i, d)
    pairs = np.array([(a, (a + j) % n) for a in range(n) for j in range(1, k + 1)] +
                     [((a + j) % n, a) for a in range(n) for j in range(1, k + 1)], capi.PAIR_DTYPE)
    off = (np.arange(len(pairs), dtype=np.uint64) * np.uint64(f))
    total = len(pairs) * f
    for r in range(repeats):
        t0 = time.perf_counter()
        ctx.match_launch(pairs, off, total)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print(f"repeat {r}: {len(pairs)} pairs x {f} x {f}: {dt * 1e3:.2f} ms, {len(pairs) * f * f / dt / 1e12:.3f} e12 distances/s")
    ctx.close()


if __name__ == "__main__":
    main()
