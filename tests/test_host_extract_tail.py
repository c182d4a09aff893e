"""CPU check of the product's host tail of extract_features (csrc/host/extract_features.cpp: strength order from the
detection order, 8 px NMS, [sparse..., dense...]) against the oracle's restatement, on synthetic keypoint lists - with
unique responses (the radix-ordered route) and with many exactly tied responses (the reference's unstable std::sort
route, where the starting order decides)."""
import numpy as np
import pytest

from opencalibration_amd import host


def _keypoints(n, seed, tied):
    rng = np.random.default_rng(seed)
    kp = np.zeros((n, 6), np.float32)
    kp[:, 0] = rng.uniform(0, 1600, n)
    kp[:, 1] = rng.uniform(0, 1200, n)
    kp[:, 5] = np.sort(rng.integers(0, 16, n))
    if tied:
        kp[:, 4] = rng.integers(1, 200, n).astype(np.float32) * np.float32(1e-4)     # ~n/200 keypoints per value
    else:
        kp[:, 4] = rng.permutation(n).astype(np.float32) * np.float32(1e-6) + np.float32(5e-5)
    desc = rng.integers(0, 2 ** 63, (n, 8), dtype=np.int64).astype(np.uint64)
    return kp, desc


@pytest.mark.parametrize("n,tied", [(0, False), (1, False), (5000, False), (5000, True), (16000, True)])
@pytest.mark.parametrize("scale", [1.0, 0.4])
def test_host_tail_matches_restatement(oracle, n, tied, scale):
    kp, desc = _keypoints(n, 100 + n + int(tied), tied)
    gloc, gst, gdesc, gns = host.extract_tail(kp, desc, scale)
    eloc, est, edesc, ens = oracle.extract_tail(kp, desc, scale)
    if n and tied:
        assert len(np.unique(est)) < n // 10
    assert gns == ens and len(gst) == len(est) == (n + 1 if n else 0)     # the seeded first keypoint appears twice
    assert np.array_equal(gst, est) and np.array_equal(gloc, eloc) and np.array_equal(gdesc, edesc)
