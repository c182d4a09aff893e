"""The bit-exact RANSAC parity (inlier sets, scores) needs device fp64 division and square root to be
correctly rounded and multiply-add to stay unfused, exactly like the reference's generic x86-64 build."""
import numpy as np
import pytest

from opencalibration_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _samples(n, seed):
    rng = np.random.default_rng(seed)
    mant = rng.uniform(1, 2, n)
    expo = rng.integers(-60, 60, n)
    x = np.ldexp(mant, expo)
    x[: n // 4] = rng.uniform(0, 1e-2, n // 4)  # the range RANSAC errors live in
    return x


def test_division_is_correctly_rounded(ctx):
    x, y = _samples(1 << 20, 1), _samples(1 << 20, 2)
    assert np.array_equal(ctx.debug_fp64(0, x, y), x / y)


def test_sqrt_is_correctly_rounded(ctx):
    x = _samples(1 << 20, 3)
    assert np.array_equal(ctx.debug_fp64(1, x), np.sqrt(x))


def test_multiply_add_is_not_fused(ctx):
    x, y = _samples(1 << 20, 4), _samples(1 << 20, 5)
    assert np.array_equal(ctx.debug_fp64(3, x, y), x * y + x)


def test_log_is_within_one_ulp(ctx):
    x = np.random.default_rng(6).uniform(1e-9, 1.0, 1 << 18)
    got, exp = ctx.debug_fp64(2, x), np.log(x)
    ulp = np.abs(got - exp) / np.spacing(np.abs(exp))
    assert ulp.max() <= 1.0
