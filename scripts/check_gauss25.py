"""OpenCV's literal gauss25 table (AKAZEFeatures.cpp, Sample_Derivative_Response_Radius6; SURF's weights) as oracle/akaze.cpp and
csrc/akaze.hip hold it, checked against what it was generated from: the Gaussian of sigma 2.5 with pi = 3.14159, printed to eight
decimals.  All 49 entries must agree (they do not with the exact pi: every entry is 7.5e-7 too large)."""
import math
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in ("oracle/akaze.cpp", "opencalibration_amd/csrc/akaze.hip"):
    text = open(os.path.join(ROOT, path)).read()
    body = text[text.index("gauss25[7][7]"):]
    vals = [float(v) for v in re.findall(r"(0\.\d{8})f", body)[:49]]
    assert len(vals) == 49, path
    for k, v in enumerate(vals):
        i, j = divmod(k, 7)
        g = math.exp(-(i * i + j * j) / 12.5) / (2 * 3.14159 * 6.25)
        assert abs(round(g, 8) - v) < 5e-9, (path, i, j, v, g)
    print(path, "gauss25: 49 of 49 entries = round(exp(-(i^2 + j^2) / 12.5) / (2 * 3.14159 * 6.25), 8)")
