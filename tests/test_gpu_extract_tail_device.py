"""extract_features' tail on the device (csrc/features.hip + csrc/std_sort.hip) against the all-host tail
(host/extract_features.cpp: extract_tail, itself checked against the oracle's restatement of
src/extract/extract_features.cpp:38-87 in tests/test_host_extract_tail.py): identical feature lists, bit for bit, on
keypoint sets with clusters (long chains of the suppression's fixed point), equal responses far apart and within the radius
of each other, a tie for the strongest feature (all of which depend on std::sort's order among equals), responses that
drive introsort to its depth limit (flagged, resolved by the host), and the empty / single-keypoint cases."""
import numpy as np
import pytest

from opencalibration_amd import capi, host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def keypoints(rng, n, w, h, clusters=0, ties=0, close_ties=0, top_tie=False):
    x = rng.uniform(2, w - 2, n).astype(np.float32)
    y = rng.uniform(2, h - 2, n).astype(np.float32)
    resp = rng.uniform(1e-4, 1.0, n).astype(np.float32)
    for c in range(clusters):            # tight clusters and lines of points 5 px apart: chains A - B - C - ...
        m = int(rng.integers(10, 60))
        i0 = int(rng.integers(0, n - m))
        cx, cy = rng.uniform(100, w - 100), rng.uniform(100, h - 100)
        if c % 2:
            x[i0:i0 + m] = cx + 5.0 * np.arange(m) * np.cos(c) % (w - 4)
            y[i0:i0 + m] = cy + 5.0 * np.arange(m) * np.sin(c) % (h - 4)
        else:
            x[i0:i0 + m] = cx + rng.normal(0, 6, m)
            y[i0:i0 + m] = cy + rng.normal(0, 6, m)
    x = np.clip(x, 0, w - 1).astype(np.float32)
    y = np.clip(y, 0, h - 1).astype(np.float32)
    for _ in range(ties):                # equal responses anywhere in the image
        a, b = rng.integers(0, n, 2)
        resp[b] = resp[a]
    for _ in range(close_ties):          # equal responses within the radius of each other
        a, b = rng.integers(0, n, 2)
        if a != b:
            resp[b] = resp[a]
            x[b] = np.float32(min(x[a] + 3.0, w - 1))
            y[b] = y[a]
    if top_tie and n > 1:
        i = np.argsort(-resp)[:2]
        resp[i[1]] = resp[i[0]]
    kp6 = np.zeros((n, 6), np.float32)
    kp6[:, 0], kp6[:, 1], kp6[:, 2], kp6[:, 4] = x, y, 4.0, resp
    desc = rng.integers(0, 2 ** 63, (n, 8), dtype=np.uint64)
    return kp6, desc


def both(ctx, kp6, desc, w, h, scale):
    lists = ctx.feature_lists(kp6, desc, (w, h), scale, subset_spacing=40.0)
    got = host.extract_tail_prepared(lists, scale)
    exp = host.extract_tail(kp6, desc, scale)
    if not lists["conflict"] and not lists["subset_conflict"]:
        # spatially_subsample_feature_indices(features, 40, num_sparse) (match_features.cpp:8-52) came with the list
        want = host.subsample(exp[0], exp[1], 40.0, exp[3])
        assert np.array_equal(lists["subset"], want), (len(lists["subset"]), len(want))
    assert got[3] == exp[3] and len(got[0]) == len(exp[0])
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2])
    forced = host.extract_tail_prepared(lists, scale, force_host_nms=True)     # the conflict path on the same lists
    assert forced[3] == exp[3] and np.array_equal(forced[0], exp[0]) and np.array_equal(forced[2], exp[2])
    return lists, exp


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_and_clustered_keypoints(ctx, seed):
    rng = np.random.default_rng(seed)
    w, h, scale = 1600, 1200, 0.4
    kp6, desc = keypoints(rng, 20000, w, h, clusters=40)
    lists, exp = both(ctx, kp6, desc, w, h, scale)
    assert not lists["conflict"] and 1000 < exp[3] < 20000
    # the device's list: both parts in descending response, the seed heading both; slot = where each keypoint went
    resp_listed = lists["records"][:, 16:20].copy().view(np.float32).ravel()
    ns = lists["num_sparse"]
    assert ns == exp[3] and np.all(np.diff(resp_listed[:ns]) <= 0) and np.all(np.diff(resp_listed[ns:]) <= 0)
    assert resp_listed[0] == resp_listed[ns] == kp6[:, 4].max()
    assert len(np.unique(lists["slot"])) == len(kp6) and lists["slot"].max() == len(kp6)


def test_equal_responses_far_apart(ctx):
    rng = np.random.default_rng(11)
    w, h, scale = 1600, 1200, 0.4
    kp6, desc = keypoints(rng, 20000, w, h, clusters=10, ties=300)
    # move tied keypoints apart so that no two equal responses interact
    order = np.argsort(kp6[:, 4], kind="stable")
    same = np.flatnonzero(np.diff(kp6[order, 4]) == 0)
    for k in same:
        a, b = order[k], order[k + 1]
        if abs(kp6[a, 0] - kp6[b, 0]) < 30 and abs(kp6[a, 1] - kp6[b, 1]) < 30:
            kp6[b, 0] = (kp6[a, 0] + 800) % (w - 1)
    lists, exp = both(ctx, kp6, desc, w, h, scale)
    assert len(same) > 100 and not lists["conflict"]


def test_equal_responses_within_the_radius(ctx):
    rng = np.random.default_rng(12)
    w, h, scale = 1600, 1200, 0.4
    kp6, desc = keypoints(rng, 15000, w, h, clusters=10, ties=50, close_ties=40)
    lists, exp = both(ctx, kp6, desc, w, h, scale)
    assert not lists["conflict"]


def test_tie_for_the_strongest(ctx):
    rng = np.random.default_rng(13)
    kp6, desc = keypoints(rng, 5000, 1600, 1200, top_tie=True)
    lists, exp = both(ctx, kp6, desc, 1600, 1200, 0.4)
    assert not lists["conflict"]


def test_depth_limit_stays_on_the_device(ctx):
    """An organ pipe of responses runs std::sort into its heap-sort fallback: restated on the device since round 5, so the
    lists come back complete and equal to the host's (until then the image was flagged and its tail redone on the host)."""
    rng = np.random.default_rng(15)
    n = 20000
    kp6, desc = keypoints(rng, n, 1600, 1200, clusters=10)
    kp6[:, 4] = (np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]) + 1.0) * 1e-4      # organ pipe
    lists, exp = both(ctx, kp6, desc, 1600, 1200, 0.4)
    assert not lists["conflict"] and not lists["subset_conflict"]


def test_small_cases(ctx):
    rng = np.random.default_rng(14)
    for n in (0, 1, 2, 7):
        kp6, desc = keypoints(rng, n, 400, 300) if n else (np.zeros((0, 6), np.float32), np.zeros((0, 8), np.uint64))
        lists, exp = both(ctx, kp6, desc, 400, 300, 1.0)
        assert len(exp[0]) == (n + 1 if n else 0)
    # every keypoint on the same pixel: one sparse feature, everything dense
    kp6, desc = keypoints(rng, 500, 400, 300)
    kp6[:, 0], kp6[:, 1] = 100.0, 100.0
    lists, exp = both(ctx, kp6, desc, 400, 300, 1.0)
    assert exp[3] == 1 and len(exp[0]) == 501
