"""Generate tests/golden/oracle_r01.json: outputs of the CPU restatement (oracle/) on fixed seeded inputs.

The reference (C++, needs Eigen/Ceres/OpenCV) cannot be built or run in this image and holds no golden vectors
of its own, so these are NOT reference outputs: they freeze the restatement, whose parity with
the reference is pinned by the restated reference tests in tests/test_oracle_*.py.  The fixture lets
`-m "not gpu"` detect any drift of the oracle (compiler, libstdc++, libm) and lets `-m gpu` check the device
against committed numbers instead of only against a checker rebuilt on the same box.

Run from the repository root:  python scripts/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc  # noqa: E402
from opencalibration_amd import synth  # noqa: E402  (data generators only)
from oracle import pyoracle  # noqa: E402


def main():
    golden = {"what": "outputs of the CPU restatement on seeded inputs (scripts/make_golden.py); not reference outputs",
              "link": [gc.oracle_link_case(pyoracle, synth, c) for c in gc.LINK_CASES],
              "extract": [gc.oracle_extract_case(pyoracle, synth, c) for c in gc.EXTRACT_CASES]}
    path = os.path.join(ROOT, "tests", "golden", "oracle_r01.json")
    with open(path, "w") as fh:
        json.dump(golden, fh, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
