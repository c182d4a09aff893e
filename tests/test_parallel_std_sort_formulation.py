"""The formulation csrc/std_sort.hip runs on the device - libstdc++'s introsort as levels of partitions, each stated from the
range's content before it (prefix ranks of the two scans' stop predicates, the pairs in order swapped) - executed in plain
Python (scripts/check_parallel_std_sort.py) against std::sort itself (och_sort_by_response(use_std=1)): the same
permutation, also among equal keys; segments that reach introsort's depth limit are reported instead.  The kernels are
held against std::sort in tests/test_gpu_std_sort.py."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_parallel_std_sort", os.path.join(ROOT, "scripts", "check_parallel_std_sort.py"))
cps = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cps)


def _same(keys):
    keys = np.asarray(keys, np.float32)
    got, fallback = cps.par_sort(keys)
    return fallback, (fallback or list(cps.std_order(keys)) == got)


@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 18, 33, 100, 1000, 3000])
def test_formulation_equals_std_sort(n):
    rng = np.random.default_rng(n + 1)
    for keys in (rng.uniform(0, 1, n), rng.integers(0, 8, n), rng.integers(0, max(n // 4, 1), n), np.arange(n), np.arange(n)[::-1],
                 np.zeros(n)):
        fallback, ok = _same(keys)
        assert ok and not fallback


def test_depth_limit_is_reported():
    fallback, ok = _same(cps.killer(2000))
    assert fallback and ok
    fallback, ok = _same(np.concatenate([np.arange(1500), np.arange(1500)[::-1]]))          # organ pipe
    assert ok
