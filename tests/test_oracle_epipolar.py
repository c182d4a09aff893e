"""the fundamental- and essential-matrix RANSAC models (tests-only in the reference).  The
restatement (oracle/epipolar.cpp) against the reference's own cases - test/test_ransac_unit.cpp:58-300 and
test/test_ransac_benchmark.cpp:263-301 with their thresholds - and its Jacobi SVD against numpy."""
import numpy as np
import pytest

from oracle import pyoracle


def _rays(points):
    """correspondence{a, b} with both measurements normalised (the unit tests' fixture loop)."""
    r = np.array([list(a) + list(b) for a, b in points], np.float64)
    r[:, :3] /= np.linalg.norm(r[:, :3], axis=1, keepdims=True)
    r[:, 3:] /= np.linalg.norm(r[:, 3:], axis=1, keepdims=True)
    return r


SQUARE8 = [((1, 2, 1),) * 2, ((2, 2, 1),) * 2, ((2, 1, 1),) * 2, ((1, 1, 1),) * 2,
           ((1, 2, 3),) * 2, ((2, 2, 2),) * 2, ((2, 1, 3),) * 2, ((1, 1, 2),) * 2]


@pytest.mark.parametrize("n", [3, 9])
def test_jacobi_svd_is_a_singular_value_decomposition(n):
    rng = np.random.default_rng(n)
    for trial in range(20):
        A = rng.normal(size=(n, n))
        if trial % 4 == 1:
            A = A @ A.T                                     # symmetric, as the 9 x 9 normal matrices are
        if trial % 4 == 2:
            A[:, -1] = A[:, 0]                              # rank deficient
        U, S, V = pyoracle.jacobi_svd(A)
        assert np.allclose(U @ np.diag(S) @ V.T, A, atol=1e-12 * max(1.0, np.abs(A).max()))
        assert np.allclose(U.T @ U, np.eye(n), atol=1e-12) and np.allclose(V.T @ V, np.eye(n), atol=1e-12)
        assert np.all(S >= 0) and np.all(np.diff(S) <= 0)
        assert np.allclose(S, np.linalg.svd(A, compute_uv=False), atol=1e-12 * max(1.0, S[0]))


@pytest.mark.parametrize("model", [0, 1])
def test_ransac_compiles(model):
    """ransac_fundamental_matrix / ransac_essential_matrix, ransac_compiles: no data, score 0, no inliers."""
    score, _, inl, _ = pyoracle.ransac_epipolar(model, np.zeros((0, 6)))
    assert score == 0 and len(inl) == 0


def test_fundamental_fits_identity():
    """test_ransac_unit.cpp:73-110."""
    rays = _rays(SQUARE8)
    score, F, inl, _ = pyoracle.ransac_epipolar(0, rays)
    assert score == pytest.approx(1.0, abs=1e-15)               # EXPECT_DOUBLE_EQ
    assert len(inl) == 8 and inl.sum() == 8
    assert np.linalg.norm(F) == pytest.approx(1.0, abs=1e-14)
    _, _, err = pyoracle.epipolar_evaluate(F, rays)
    assert err.sum() == pytest.approx(0.0, abs=1e-10)


OUTLIER_MIX = [((1, 2, 1), (1, 2, 1)), ((100, 200, 1), (200, 100, 1)), ((2, 2, 1), (2, 2, 1)), ((150, 250, 1), (250, 150, 1)),
               ((2, 1, 1), (2, 1, 1)), ((1, 1, 1), (1, 1, 1)), ((1.5, 1.5, 1), (1.5, 1.5, 1)), ((120, 220, 1), (220, 120, 1)),
               ((1.2, 1.8, 1), (1.2, 1.8, 1)), ((130, 230, 1), (230, 130, 1)), ((1.8, 1.2, 1), (1.8, 1.2, 1)),
               ((1.3, 1.7, 1), (1.3, 1.7, 1)), ((1, 2, 3), (1, 2, 3)), ((2, 2, 2), (2, 2, 2))]
OUTLIER_MIX_INLIERS = [True, False, True, False, True, True, True, False, True, False, True, True, True, True]


@pytest.mark.parametrize("model", [0, 1])
def test_fit_inliers_uses_correct_subset(model):
    """test_ransac_unit.cpp:178-232 (fundamental) and :300-347 (essential): the flagged subset alone shapes the model."""
    rays = _rays(OUTLIER_MIX)
    inl = np.array(OUTLIER_MIX_INLIERS)
    M = pyoracle.epipolar_fit_inliers(model, rays, inl)
    _, _, err = pyoracle.epipolar_evaluate(M, rays)
    avg_in, avg_out = np.abs(err[inl]).mean(), np.abs(err[~inl]).mean()
    assert avg_in < 0.01 and avg_out > 2 * avg_in


def test_fundamental_evaluate_uses_absolute_error():
    """test_ransac_unit.cpp:234-259."""
    pts = [((1, 2, 1),) * 2, ((2, 2, 1),) * 2, ((2, 1, 1),) * 2, ((1, 1, 1),) * 2, ((1.5, 1.5, 1),) * 2,
           ((1.2, 1.8, 1),) * 2, ((1.8, 1.2, 1),) * 2, ((1.3, 1.7, 1),) * 2, ((1, 2, 3),) * 2, ((2, 2, 2),) * 2]
    score, _, inl, _ = pyoracle.ransac_epipolar(0, _rays(pts))
    assert score > 0.7 and inl.sum() >= 8


def test_essential_fits_identity():
    """test_ransac_unit.cpp:271-298."""
    pts = [((1, 2, 1),) * 2, ((2, 2, 1),) * 2, ((2, 1, 1),) * 2, ((1, 1, 1),) * 2, ((1, 2, 3),) * 2, ((2, 2, 2),) * 2]
    score, E, inl, _ = pyoracle.ransac_epipolar(1, _rays(pts))
    assert score >= 0.16 and len(inl) == 6 and inl.sum() >= 1
    s = np.linalg.svd(E, compute_uv=False)
    assert s[0] == pytest.approx(s[1], rel=1e-9) and s[2] == pytest.approx(0.0, abs=1e-12 * s[0])   # an essential matrix


def test_essential_decomposition_recovers_the_motion():
    """essential_matrix_model::decompose (:130-153): E = [t]_x R yields +-t and two rotations, one of them R."""
    ang = 0.2
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([0.6, 0.0, 0.8])
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    ok, poses = pyoracle.essential_decompose(tx @ R)
    assert ok
    for p in poses:
        assert np.linalg.norm(p[:4]) == pytest.approx(1.0, abs=1e-12)
        assert abs(abs(p[4:] @ t) - 1.0) < 1e-9                                 # translation = +-t
    def rot(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    assert min(np.abs(rot(p[:4]) - R).max() for p in poses) < 1e-9
    assert np.array_equal(poses[0][:4], poses[1][:4]) and np.array_equal(poses[0][4:], -poses[1][4:])


def _precision_recall(inl, gt):
    tp, fp, fn = (inl & gt).sum(), (inl & ~gt).sum(), (~inl & gt).sum()
    return (tp / (tp + fp) if tp else 0.0), (tp / (tp + fn) if tp else 0.0)


@pytest.mark.parametrize("n_in,n_out,planar,prec,rec", [(200, 0, 0.0, 0.95, 0.80), (140, 60, 0.0, 0.85, 0.70), (200, 0, 0.8, 0.95, 0.95)])
def test_fundamental_benchmarks(n_in, n_out, planar, prec, rec):
    """test_ransac_benchmark.cpp:263-301: fundamental_clean, fundamental_30pct_outliers, fundamental_dominant_plane (the
    DEGENSAC case: 80 % of the inliers on one plane)."""
    corr, gt, F_gt = pyoracle.scene_fundamental(n_in, n_out, planar, 42)
    score, F, inl, iterations = pyoracle.ransac_epipolar(0, corr[:, :6])
    p, r = _precision_recall(inl, gt.astype(bool))
    assert p >= prec and r >= rec, (p, r, iterations)
    if n_out == 0 and planar == 0.0:
        # the model's fit solves p1^T F p2 = 0 (fundamental_matrix_model.cpp:55, rows x * x_, x * y_, x, ...) while its
        # error() and the benchmark's ground truth read x2^T F x1: what comes out is the TRANSPOSE of the ground truth
        # (the benchmark never looks at the matrix of this model, only at precision and recall)
        Fn = F / np.linalg.norm(F)
        assert min(np.linalg.norm(Fn - F_gt.T), np.linalg.norm(Fn + F_gt.T)) < 1e-3
