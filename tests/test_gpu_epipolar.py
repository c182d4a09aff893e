"""SURVEY.md section 8 row a9 on the device: ransac<fundamental_matrix_model> / ransac<essential_matrix_model>
(csrc/ransac.hip: ransac_epipolar_kernel - 8-point / 5-point fits through a restated JacobiSVD, Sampson error, DEGENSAC,
the shared RANSAC loop) against the oracle's restatement bit for bit - inlier sets, iteration counts, scores, matrices - on
the reference's unit-test fixtures (test/test_ransac_unit.cpp:58-347) and benchmark scenes (test_ransac_benchmark.cpp:263-301),
and at the reference's own thresholds."""
import numpy as np
import pytest

from opencalibration_amd import capi, host
from test_oracle_epipolar import OUTLIER_MIX, SQUARE8, _precision_recall, _rays

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _same(ctx, oracle, model, rays, quality=None, threshold=0.01):
    es, eM, einl, eit = oracle.ransac_epipolar(model, rays, quality, threshold)
    gs, gM, ginl, git, _ = host.ransac_epipolar(ctx, model, rays, quality, threshold)
    assert git == eit and np.array_equal(ginl, einl), (git, eit, ginl.sum(), einl.sum())
    assert gs == es and np.array_equal(gM, eM, equal_nan=True)
    return gs, gM, ginl, git


@pytest.mark.parametrize("model", [0, 1])
def test_no_data(ctx, oracle, model):
    score, M, inl, it, _ = host.ransac_epipolar(ctx, model, np.zeros((0, 6)))
    assert score == 0 and len(inl) == 0
    score, M, inl, it, _ = host.ransac_epipolar(ctx, model, _rays(SQUARE8)[:4])      # fewer than MINIMUM_POINTS
    assert score == 0 and not inl.any() and np.isnan(M).all()


def test_fundamental_fits_identity(ctx, oracle):
    """test_ransac_unit.cpp:73-110"""
    rays = _rays(SQUARE8)
    score, F, inl, _ = _same(ctx, oracle, 0, rays)
    assert score == pytest.approx(1.0, abs=1e-15) and inl.sum() == 8
    assert np.linalg.norm(F) == pytest.approx(1.0, abs=1e-14)


def test_fundamental_with_outliers_and_essential(ctx, oracle):
    """:112-176 / :234-298"""
    rays = _rays(OUTLIER_MIX)
    score, F, inl, _ = _same(ctx, oracle, 0, rays)
    assert inl.sum() >= 8
    pts = [((1, 2, 1),) * 2, ((2, 2, 1),) * 2, ((2, 1, 1),) * 2, ((1, 1, 1),) * 2, ((1, 2, 3),) * 2, ((2, 2, 2),) * 2]
    score, E, inl, _ = _same(ctx, oracle, 1, _rays(pts))
    assert score >= 0.16 and inl.sum() >= 1
    s = np.linalg.svd(E, compute_uv=False)
    assert s[0] == pytest.approx(s[1], rel=1e-9) and s[2] == pytest.approx(0.0, abs=1e-12 * s[0])
    _same(ctx, oracle, 1, rays)


@pytest.mark.parametrize("n_in,n_out,planar,prec,rec", [(200, 0, 0.0, 0.95, 0.80), (140, 60, 0.0, 0.85, 0.70), (200, 0, 0.8, 0.95, 0.95)])
def test_fundamental_benchmarks(ctx, oracle, n_in, n_out, planar, prec, rec):
    """test_ransac_benchmark.cpp:263-301: clean, 30 % outliers, dominant plane (DEGENSAC)"""
    corr, gt, F_gt = oracle.scene_fundamental(n_in, n_out, planar, 42)
    score, F, inl, iterations = _same(ctx, oracle, 0, corr[:, :6])
    p, r = _precision_recall(inl, gt.astype(bool))
    assert p >= prec and r >= rec, (p, r, iterations)
    # with match qualities the sampling is PROSAC's
    score, F, inl, iterations = _same(ctx, oracle, 0, corr[:, :6], quality=corr[:, 6] if corr.shape[1] > 6 else None)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_scenes_with_quality(ctx, oracle, seed):
    """general two-view scenes with outliers and match qualities (PROSAC order with ties): both models, long runs"""
    rng = np.random.default_rng(seed)
    n_in, n_out = 150, 90
    X = np.concatenate([rng.uniform(-2, 2, (n_in, 2)), rng.uniform(4, 9, (n_in, 1))], axis=1)
    ang = 0.15
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([0.7, 0.1, 0.05])
    x1 = X / X[:, 2:]
    X2 = X @ R.T + t
    x2 = X2 / X2[:, 2:]
    x1[:, :2] += rng.normal(0, 1e-3, (n_in, 2))
    x2[:, :2] += rng.normal(0, 1e-3, (n_in, 2))
    o1 = np.concatenate([rng.uniform(-0.5, 0.5, (n_out, 2)), np.ones((n_out, 1))], axis=1)
    o2 = np.concatenate([rng.uniform(-0.5, 0.5, (n_out, 2)), np.ones((n_out, 1))], axis=1)
    rays = np.concatenate([np.concatenate([x1, x2], axis=1), np.concatenate([o1, o2], axis=1)])
    rays = rays[rng.permutation(len(rays))]
    quality = rng.integers(1, 60, len(rays)) / 486.0            # many ties, as Hamming distances make them
    for model in (0, 1):
        _same(ctx, oracle, model, rays)
        s, M, inl, it = _same(ctx, oracle, model, rays, quality)
        assert it >= 20
