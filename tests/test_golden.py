"""Committed fixtures (tests/golden/oracle_r01.json): the oracle must reproduce them bit for bit on any box (CPU
suite), and so must the device path (GPU suite): match lists, inlier sets, RANSAC scores and homographies of every
directed pair of two small surveys (one with lens distortion), AKAZE keypoints / descriptors and the extract_features
output of three rendered images."""
import json
import os

import pytest

import golden_cases as gc
from opencalibration_amd import synth

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_r01.json")))


@pytest.mark.parametrize("k", range(len(gc.LINK_CASES)))
def test_oracle_link_matches_golden(oracle, k):
    assert gc.oracle_link_case(oracle, synth, gc.LINK_CASES[k]) == GOLDEN["link"][k]


@pytest.mark.parametrize("k", range(len(gc.EXTRACT_CASES)))
def test_oracle_extract_matches_golden(oracle, k):
    assert gc.oracle_extract_case(oracle, synth, gc.EXTRACT_CASES[k]) == GOLDEN["extract"][k]


@pytest.fixture(scope="module")
def ctx():
    from opencalibration_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(gc.LINK_CASES)))
def test_device_link_matches_golden(ctx, k):
    from opencalibration_amd import host

    got, exp = gc.device_link_case(ctx, host, synth, gc.LINK_CASES[k]), GOLDEN["link"][k]
    assert got["subsets"] == exp["subsets"] and len(got["pairs"]) == len(exp["pairs"])
    for g, e in zip(got["pairs"], exp["pairs"]):
        assert g == e, (g["a"], g["b"])


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(gc.EXTRACT_CASES)))
def test_device_extract_matches_golden(ctx, k):
    from opencalibration_amd import host

    assert gc.device_extract_case(ctx, host, synth, gc.EXTRACT_CASES[k]) == GOLDEN["extract"][k]
