import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The pins against the reference's own headers (oracle/_ref/libref.so, built in place from /root/reference) must not
    # vanish silently: where the reference is mounted they are REQUIRED (a missing library fails the tests that use it);
    # elsewhere - the GPU box, a clone - the prebuilt library travels with the tree, and if it is absent the tests skip
    # and the count is printed at the end of the run.
    if os.path.isdir("/root/reference/external/jk-tree/include"):
        os.environ.setdefault("OCHIP_REQUIRE_REF", "1")


def require_ref(oracle, symbol=None):
    """oracle/_ref/libref.so for a pin test: fails under OCHIP_REQUIRE_REF=1 when it is missing, skips (counted) otherwise."""
    r = oracle.ref()
    if r is not None and (symbol is None or hasattr(r, symbol)):
        return r
    msg = "oracle/_ref/libref.so is not built" + ("" if r is None else f" with {symbol}") + " (make -C oracle ref, needs /root/reference)"
    if os.environ.get("OCHIP_REQUIRE_REF") == "1":
        pytest.fail(msg + " and OCHIP_REQUIRE_REF=1: the pins against the reference's own headers are required here")
    _REF_SKIPS.append(msg)
    pytest.skip(msg)


_REF_SKIPS = []


def pytest_terminal_summary(terminalreporter):
    if _REF_SKIPS:
        terminalreporter.write_line(f"reference pins SKIPPED: {len(_REF_SKIPS)} test(s) could not load oracle/_ref/libref.so "
                                    "(KD-tree, GridFilter, UnionFind, KMeans, Hilbert pins did not run)", yellow=True)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.lib()
    return pyoracle
