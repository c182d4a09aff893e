"""GPU parity of the Hamming 2-NN kernel (ochip_match_batch) against the oracle, through the C ABI.
Bar: bit-exact (feature_index_1, feature_index_2, popcount) and identical post-sort order."""
import numpy as np
import pytest

from opencalibration_amd import capi, host, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _upload(ctx, descs):
    ctx.descriptors_reserve(len(descs), sum(len(d) for d in descs))
    for i, d in enumerate(descs):
        ctx.upload_descriptors(i, d)


def _popcount(a):
    return np.unpackbits(np.ascontiguousarray(a).view(np.uint8), axis=-1).sum(-1)


def _expect(d1, d2):
    ham = _popcount(d1[:, None, :] ^ d2[None, :, :]).astype(np.int64)
    key = ham * (1 << 20) + np.arange(d2.shape[0])[None, :]
    order = np.argsort(key, axis=1, kind="stable")
    best_k = order[:, 0]
    best = ham[np.arange(len(d1)), best_k]
    second = ham[np.arange(len(d1)), order[:, 1]] if d2.shape[0] > 1 else np.full(len(d1), capi.NO_SECOND)
    return best_k, best, second


@pytest.mark.parametrize("n1,n2", [(1, 1), (1, 2), (3, 1), (64, 65), (257, 511), (513, 300), (1000, 1025)])
def test_raw_top2_against_numpy(ctx, n1, n2):
    rng = np.random.default_rng(n1 * 7919 + n2)
    base = synth.descriptors_for_ids(np.arange(max(n1, n2)))
    d1, d2 = base[:n1].copy(), base[:n2].copy()
    for d in (d1, d2):
        bits = rng.integers(0, 486, (len(d), 30))
        for j in range(30):
            d[np.arange(len(d)), bits[:, j] >> 6] ^= np.uint64(1) << (bits[:, j] & 63).astype(np.uint64)
    if n2 > 4:
        d2[3] = d2[1]  # exact duplicates: ties must resolve to the lowest k and second == best
    _upload(ctx, [d1, d2])
    pairs = np.array([(0, 1), (1, 0)], capi.PAIR_DTYPE)
    out = ctx.match_batch(pairs, np.array([0, n1], np.uint64), n1 + n2)
    for (a, b, off) in ((d1, d2, 0), (d2, d1, n1)):
        bk, bc, sc = _expect(a, b)
        got = out[off:off + len(a)]
        assert np.array_equal(got["best_k"], bk)
        assert np.array_equal(got["best_count"], bc)
        assert np.array_equal(got["second_count"], sc)


def test_mixed_batch_of_paired_and_unpaired_directions(ctx):
    """Pairs whose reverse is in the batch take the both-directions-from-one-pass kernel, the others the one-direction
    kernel; a duplicate of a pair is matched on its own.  Every record must be the brute-force answer either way."""
    rng = np.random.default_rng(77)
    base = synth.descriptors_for_ids(np.arange(700))
    sizes = [300, 517, 64, 129]
    descs = []
    for n in sizes:
        d = base[rng.permutation(700)[:n]].copy()
        bits = rng.integers(0, 486, (n, 25))
        for j in range(25):
            d[np.arange(n), bits[:, j] >> 6] ^= np.uint64(1) << (bits[:, j] & 63).astype(np.uint64)
        descs.append(d)
    _upload(ctx, descs)
    plist = [(0, 1), (1, 0), (0, 2), (3, 1), (1, 3), (2, 3), (0, 1), (2, 2)]
    pairs = np.array(plist, capi.PAIR_DTYPE)
    off = np.cumsum([0] + [sizes[a] for a, _ in plist]).astype(np.uint64)
    out = ctx.match_batch(pairs, off[:-1].copy(), int(off[-1]))
    for p, (a, b) in enumerate(plist):
        bk, bc, sc = _expect(descs[a], descs[b])
        got = out[int(off[p]):int(off[p + 1])]
        assert np.array_equal(got["best_k"], bk) and np.array_equal(got["best_count"], bc) and np.array_equal(got["second_count"], sc), (a, b)


def test_empty_reference_set_and_empty_batch(ctx):
    d1 = synth.descriptors_for_ids(np.arange(10))
    _upload(ctx, [d1, np.zeros((0, 8), np.uint64)])
    out = ctx.match_batch(np.array([(0, 1), (1, 0)], capi.PAIR_DTYPE), np.array([0, 10], np.uint64), 10)
    m = host.matches_from_device(out, np.arange(10, dtype=np.uint64), np.zeros(0, np.uint64))
    assert len(m[0]) == 0
    assert len(ctx.match_batch(np.zeros(0, capi.PAIR_DTYPE), np.zeros(0, np.uint64), 0)) == 0


def test_errors_are_reported(ctx):
    _upload(ctx, [synth.descriptors_for_ids(np.arange(4))])
    with pytest.raises(capi.OchipError):
        ctx.upload_descriptors(0, synth.descriptors_for_ids(np.arange(4)))  # twice
    with pytest.raises(capi.OchipError):
        ctx.match_batch(np.array([(0, 5)], capi.PAIR_DTYPE), np.array([0], np.uint64), 4)  # unknown image


def test_match_features_subset_parity_on_synthetic_grid(ctx, oracle):
    """Full match_features_subset semantics (ratio test, remap, std::sort order) vs the oracle on a
    C1-shaped grid: every directed kNN pair, bit-exact triples in identical order."""
    g = synth.make_grid(2, 4, feats=1024, seed=9)
    subsets = [oracle.subsample(*g.image(i)[:2], 40.0, int(g.num_sparse[i])) for i in range(g.n_images)]
    _upload(ctx, [g.image(i)[2][subsets[i].astype(np.int64)] for i in range(g.n_images)])
    pairs = np.array([(a, b) for a in range(g.n_images) for b in range(g.n_images) if a != b], capi.PAIR_DTYPE)
    n1 = np.array([len(subsets[a]) for a in pairs["image_1"]], np.uint64)
    off = np.concatenate([[0], np.cumsum(n1)[:-1]]).astype(np.uint64)
    out = ctx.match_batch(pairs, off, int(n1.sum()))
    for p, (a, b) in enumerate(pairs):
        raw = out[int(off[p]):int(off[p] + n1[p])]
        i1, i2, dist = host.matches_from_device(raw, subsets[a], subsets[b])
        e1, e2, ed = oracle.match(g.image(a)[2], g.image(b)[2], subsets[a], subsets[b])
        assert np.array_equal(i1, e1) and np.array_equal(i2, e2) and np.array_equal(dist, ed)


def test_full_size_properties(ctx):
    """BASELINE-size (4k x 4k) properties that need no oracle: planted twins are found with the exact
    planted distance, d(a,a) = 0 with itself as best, symmetry of the best distance."""
    rng = np.random.default_rng(1)
    n = 4096
    d1 = synth.descriptors_for_ids(np.arange(n) + 10_000)
    d2 = synth.descriptors_for_ids(np.arange(n) + 10_000)[rng.permutation(n)]
    flips = rng.integers(1, 60, n)
    d2n = d2.copy()
    for i in range(n):
        bits = rng.choice(486, flips[i], replace=False)
        for b in bits:
            d2n[i, b >> 6] ^= np.uint64(1) << np.uint64(b & 63)
    _upload(ctx, [d1, d2n, d1.copy()])
    pairs = np.array([(1, 0), (0, 2)], capi.PAIR_DTYPE)
    out = ctx.match_batch(pairs, np.array([0, n], np.uint64), 2 * n)
    a = out[:n]
    # twin of d2n[i] is the d1 row with the same id: distance = number of flipped bits
    twin = np.array([np.flatnonzero((d1 == d2[i]).all(1))[0] for i in range(0, n, 64)])
    assert np.array_equal(a["best_k"][::64], twin)
    assert np.array_equal(a["best_count"], flips)
    b = out[n:]
    assert np.array_equal(b["best_k"], np.arange(n)) and np.all(b["best_count"] == 0)
    assert np.all(b["second_count"] > 150)


def test_parity_holds_when_the_partials_do_not_fit():
    """OCHIP_MATCH_SYM_CAP_MB=1: the column partials of the both-directions kernel are limited to 1 MB in flight, so the
    paired jobs of a launch run in many groups that re-use the buffer (and a pair larger than the cap goes one direction
    at a time).  The switch is read once per process: this file's parity tests are re-run in a child with it set."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OCHIP_MATCH_SYM_CAP_MB="1", OCHIP_TEST_HOOKS="popcount_match")   # (the popcount kernels: see the last test)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_match.py"), os.path.join(root, "tests", "test_gpu_link.py"),
                        "-x", "-q", "-m", "gpu", "-k", "not partials_do_not_fit and not popcount_kernels"], cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("n1,n2", [(100, 8192), (70, 8193)])
def test_reference_sets_at_the_matrix_core_kernels_limit(ctx, n1, n2):
    """The matrix-core kernel carries the reference index in 13 bits of an fp32 key: 8 192 references are its largest
    set, one more goes to the popcount kernels.  Both sides of the limit must give the brute-force answer."""
    rng = np.random.default_rng(n2)
    d2 = synth.descriptors_for_ids(np.arange(n2) + 50_000)
    d1 = d2[rng.permutation(n2)[:n1]].copy()
    bits = rng.integers(0, 486, (n1, 12))
    for j in range(12):
        d1[np.arange(n1), bits[:, j] >> 6] ^= np.uint64(1) << (bits[:, j] & 63).astype(np.uint64)
    d2[n2 - 1] = d2[n2 - 2]     # a tie between the last two references
    _upload(ctx, [d1, d2])
    out = ctx.match_batch(np.array([(0, 1)], capi.PAIR_DTYPE), np.array([0], np.uint64), n1)
    bk, bc, sc = _expect(d1, d2)
    assert np.array_equal(out["best_k"], bk) and np.array_equal(out["best_count"], bc) and np.array_equal(out["second_count"], sc)


def test_popcount_kernels_still_agree():
    """OCHIP_TEST_HOOKS=popcount_match sends every pair to the popcount kernels (the path of images with more than 8 192 features in
    the matcher).  The switch is read once per process: this file's parity tests are re-run in a child with it set."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, OCHIP_TEST_HOOKS="popcount_match")
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-m", "gpu", "-k",
                        "raw_top2 or mixed_batch or full_size or subset_parity"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
