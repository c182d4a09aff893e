"""The greedy suppression of extract_features (extract_features.cpp:58-83) and of spatially_subsample_feature_indices
(match_features.cpp:8-52) as the FIXED POINT csrc/features.hip iterates: a point is kept when every stronger point within
the radius is suppressed, suppressed as soon as one of them is kept.  Plain Python against the sequential greedy pass on
clustered points: same kept set, reached in as many rounds as the longest chain of undecided neighbours.  The kernels are
held against the host's greedy pass in tests/test_gpu_extract_tail_device.py."""
import numpy as np


def greedy(xy, radius):
    kept = []
    state = np.zeros(len(xy), np.int8)
    for i, p in enumerate(xy):
        near = False
        for k in kept:
            d = xy[k] - p
            if not (d[0] * d[0] + d[1] * d[1] > radius * radius):
                near = True
                break
        if not near:
            kept.append(i)
            state[i] = 1
        else:
            state[i] = 2
    return state


def fixed_point(xy, radius):
    n = len(xy)
    d2 = ((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1)
    within = ~(d2 > radius * radius)
    stronger = np.tril(np.ones((n, n), bool), -1)               # [i, q]: q stronger than i (q < i)
    nb = within & stronger
    state = np.zeros(n, np.int8)                                # 0 undecided, 1 kept, 2 suppressed
    rounds = 0
    while (state == 0).any():
        rounds += 1
        prev = state.copy()                                     # (a round reads the states of the round before)
        for i in np.flatnonzero(prev == 0):
            q = np.flatnonzero(nb[i])
            if (prev[q] == 1).any():
                state[i] = 2
            elif not (prev[q] == 0).any():
                state[i] = 1
        assert rounds <= n
    return state, rounds


def test_fixed_point_is_the_greedy_pass():
    rng = np.random.default_rng(3)
    for trial in range(4):
        pts = [rng.uniform(0, 400, (300, 2))]
        for c in range(12):                                      # clusters and chains 5 px apart
            m = int(rng.integers(5, 40))
            base = rng.uniform(50, 350, 2)
            if c % 2:
                pts.append(base + np.outer(np.arange(m), [5.0, 0.7]))
            else:
                pts.append(base + rng.normal(0, 6, (m, 2)))
        xy = rng.permutation(np.concatenate(pts))                # order = strength order
        want = greedy(xy, 8.0)
        got, rounds = fixed_point(xy, 8.0)
        assert np.array_equal(got, want)
        assert got[0] == 1 and 1 < rounds < 60
