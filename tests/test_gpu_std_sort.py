"""libstdc++'s std::sort on the device (csrc/std_sort.hip) against std::sort itself (och_sort_by_response(use_std=1) calls
it on (response, index) records with comp = response greater): the permutation must be identical, also among equal keys and
for segments that hit introsort's depth limit (heap sort)."""
import ctypes as C

import numpy as np
import pytest

from opencalibration_amd import capi, host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def std_order(keys_u32):
    """std::sort by descending key from the order 0..n-1 (keys as float bit patterns of positive floats: same order)."""
    L = host.load()
    r = np.ascontiguousarray(keys_u32, np.uint32).view(np.float32)
    out = np.zeros(max(len(r), 1), np.uint32)
    L.och_sort_by_response.restype = None
    L.och_sort_by_response.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
    L.och_sort_by_response(r.ctypes.data, len(r), out.ctypes.data, 1)
    return out[:len(r)]


def as_keys(values):
    """non-negative numbers -> uint32 keys that order like them and are valid positive-float bit patterns"""
    v = np.asarray(values, np.float64)
    return (np.asarray(v + 1.0, np.float32)).view(np.uint32).copy()


def killer(n):
    k = n // 2
    a = np.zeros(n)
    for i in range(1, k + 1):
        if i % 2 == 1:
            a[i - 1] = i
            a[i] = k + i
        a[k + i - 1] = 2 * i
    return a


def run(ctx, segments):
    keys = np.concatenate([as_keys(s) for s in segments]) if segments else np.zeros(0, np.uint32)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in segments])]).astype(np.uint32)
    payload = np.concatenate([np.arange(len(s), dtype=np.uint32) for s in segments]) if segments else np.zeros(0, np.uint32)
    ko, po, fb = ctx.std_sort(keys, payload, offsets)
    flagged = 0
    for s, seg in enumerate(segments):
        a, b = int(offsets[s]), int(offsets[s + 1])
        if fb[s]:
            flagged += 1
            assert sorted(po[a:b].tolist()) == list(range(b - a))        # still a permutation of the input
            continue
        exp = std_order(keys[a:b])
        assert np.array_equal(po[a:b], exp), (s, len(seg))
        assert np.array_equal(ko[a:b], keys[a:b][exp])
    return flagged


def test_sizes_around_the_threshold_and_patterns(ctx):
    rng = np.random.default_rng(1)
    segs = []
    for n in (0, 1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 64, 65, 100, 257, 1000, 4095, 4096, 4097, 5000, 9000):
        segs += [rng.uniform(0, 1, n), rng.integers(0, 8, n), rng.integers(0, max(n // 4, 1), n), np.arange(n), np.arange(n)[::-1],
                 np.zeros(n)]
    assert run(ctx, segs) == 0


def test_keypoint_sized_segments_with_ties(ctx):
    rng = np.random.default_rng(2)
    segs = []
    for i in range(24):
        n = int(rng.integers(15000, 30000))
        r = rng.uniform(1e-4, 1, n)
        for _ in range(int(rng.integers(0, 40))):
            a, b = rng.integers(0, n, 2)
            r[b] = r[a]
        segs.append(r)
    assert run(ctx, segs) == 0


def test_match_sized_segments_with_heavy_ties(ctx):
    rng = np.random.default_rng(3)
    # Hamming counts: a few dozen distinct values over ~1 200 matches; then the PROSAC order's input: the same counts
    # already in descending order, sorted ascending (complemented keys)
    segs = [rng.integers(20, 120, int(rng.integers(0, 3500))) for _ in range(300)]
    segs += [1000 - np.sort(rng.integers(20, 120, int(rng.integers(0, 3500))))[::-1] for _ in range(300)]
    assert run(ctx, segs) == 0


def test_depth_limit_heap_sort_on_the_device(ctx):
    """Median-of-three killers and organ pipes run introsort into its depth limit, where libstdc++ heap-sorts the range
    (std::__partial_sort): restated on the device since round 5 (one lane, sequential) - nothing is flagged for the host any
    more and the permutation is std::sort's, for ranges that end in LDS (killer(200), killer(3000)), for long ones
    (killer(20000): heap-sorted through LDS; killer(40000): in place) and with ties."""
    rng = np.random.default_rng(4)
    tied = np.floor(killer(6000) / 3)
    segs = [killer(200), -killer(200) + 1000, killer(3000), killer(20000), killer(40000), tied,
            np.concatenate([np.arange(10000), np.arange(10477)[::-1]]), rng.uniform(0, 1, 5000)]
    assert run(ctx, segs) == 0


def test_many_segments_take_the_workgroup_per_segment_route(ctx):
    """32 or more segments: one workgroup per segment walks the levels above 1 024 records by itself (sort_segments_kernel);
    fewer take a launch per level - both against std::sort, killers among them."""
    rng = np.random.default_rng(6)
    segs = [rng.uniform(0, 1, int(rng.integers(900, 26000))) for _ in range(40)] + [killer(20000), rng.integers(0, 50, 30000)]
    assert run(ctx, segs) == 0
    assert run(ctx, segs[:5] + [killer(20000)]) == 0


def test_both_routes_on_the_same_segments(ctx, monkeypatch):
    """The workgroup-per-segment route (from 8 segments on: a from-host chunk of 25 images takes it too) and the launch-per-level
    route (OCHIP_TEST_HOOKS=sort_per_level, read per call) on the same segments: both are libstdc++'s permutation."""
    rng = np.random.default_rng(23)
    segs = [rng.integers(0, 4000, int(rng.integers(15000, 26000))) for _ in range(25)]       # 25 images' keypoints, tied responses
    assert run(ctx, segs) == 0
    assert run(ctx, segs[:8]) == 0 and run(ctx, segs[:7]) == 0                                 # either side of the threshold
    monkeypatch.setenv("OCHIP_TEST_HOOKS", "sort_per_level")
    assert run(ctx, segs) == 0
