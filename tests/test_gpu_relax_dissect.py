"""The dissected factorisation (csrc/relax.hip, assign_tangent: regions of the camera graph + separators in the dense
tail, regions factored side by side, one back-substitution workgroup per region) against the single band: the same
LM trajectory on a 320-camera survey, and the tile factorisation against the launch chain under the new ordering."""
import os
import subprocess
import sys

import numpy as np
import pytest

from opencalibration_amd import capi, host, pipeline, synth
from relax_fixtures import qangle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dissected_relax_equals_the_single_band(monkeypatch):
    grid = synth.make_grid(seed=5, rows=16, cols=20, feats=512)
    start = pipeline.perturbed_orientations(grid, 0.1, 4)
    out, mem = {}, {}
    for mode in ("dissected", "band"):
        if mode == "band":
            monkeypatch.setenv("OCHIP_TEST_HOOKS", "no_dissect")
        ctx = capi.Context(0)
        g = host.Graph.from_synthetic(grid)
        g.link(ctx)
        g.set_orientations(start)
        out[mode] = g.relax_ground_plane(ctx, start)
        mem[mode] = ctx.relax_memory()
        g.close()
        ctx.close()
    a, b = out["dissected"], out["band"]
    assert int(a["iterations_total"]) == int(b["iterations_total"]) and int(a["residual_blocks"]) == int(b["residual_blocks"])
    worst = max(qangle(a["orientation"][i], b["orientation"][i]) for i in range(grid.n_images))
    assert worst < 1e-9, worst
    assert abs(a["final_cost"] - b["final_cost"]) <= 1e-9 * abs(b["final_cost"])
    # the separators' dense rows are stored under every column: the dissection was in effect
    assert mem["dissected"][0] == mem["band"][0] == 3 * grid.n_images + 3
    assert mem["dissected"][1] > mem["band"][1]
    err = max(qangle(a["orientation"][i], grid.orientation[i]) for i in range(grid.n_images))
    assert err < 5e-3, err


@pytest.mark.parametrize("knobs", [{}, {"OCHIP_RELAX_DISSECT_G": "64"}, {"OCHIP_TEST_HOOKS": "no_dissect"}],
                         ids=lambda k: "+".join("%s=%s" % kv for kv in k.items()) or "default")
def test_tile_factorisation_equals_the_chain_under_the_dissection(knobs):
    """OCHIP_TEST_HOOKS=chol_verify factors every system of the solve both ways (one launch of tiles / the launch chain) and fails
    the relax when the forward solves differ by more than 1e-7 relative; the knobs are read once per process, so every
    ordering (default regions, small regions, one band) runs in a process of its own.  (The kernel's round-3 A/B variants -
    unfused pairs, whole-tile operands, rank-1 diagonal tiles, plain summation order - lost and are gone.)"""
    env = dict(os.environ, OCHIP_VERBOSE="relax", **knobs)
    env["OCHIP_TEST_HOOKS"] = ",".join(["chol_verify"] + ([knobs["OCHIP_TEST_HOOKS"]] if "OCHIP_TEST_HOOKS" in knobs else []))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "probe_relax_dissect.py"), "16x20x512", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "factorisation check" in r.stderr and "LM it/s" in r.stdout
    regions = [l for l in r.stderr.splitlines() if "regions" in l and "n=963" in l]
    assert regions
    if knobs.get("OCHIP_TEST_HOOKS") == "no_dissect":
        assert " 1 regions" in regions[0], regions
    else:
        assert " 1 regions" not in regions[0], regions


def _probe(knobs):
    env = dict(os.environ, **knobs)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "probe_relax_dissect.py"), "16x20x512", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("final_cost")][0].split()
    iterations = int([l for l in r.stdout.splitlines() if "iterations" in l][0].split(":")[1].split()[0])
    return iterations, float(line[1]), float(line[3])


def test_back_substitution_variants_agree():
    """The backward substitution with its part of x in LDS or in HBM (the fallback for systems that do not fit), dissected
    and as one band: same LM trajectory."""
    base = _probe({})
    hbm = _probe({"OCHIP_TEST_HOOKS": "back_solve_x_global"})
    assert hbm[0] == base[0] and hbm[1] == base[1] and hbm[2] == base[2]        # same arithmetic: same bits
    band = _probe({"OCHIP_TEST_HOOKS": "no_dissect"})
    assert band[0] == base[0]
    assert abs(band[1] - base[1]) <= 1e-9 * abs(base[1]) and abs(band[2] - base[2]) <= 1e-9 * abs(base[2])


def _mesh_probe(extra):
    env = dict(os.environ, OCHIP_VERBOSE="relax", **extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "probe_relax_mesh.py"), "C3", "256"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = {}
    for line in r.stdout.splitlines():
        for name in ("mesh, 4 vertices", "mesh, grid"):
            if line.startswith(name):
                d = eval(line[line.index("{"):line.index("}") + 1])
                rows[name] = (int(d["iterations_total"]), float(d["final_cost"]))
    regions = [int(l.split(";")[1].split()[0]) for l in r.stderr.splitlines() if "[ochip relaxg]" in l and "regions" in l]
    return rows, max(regions) if regions else 0


def test_general_engine_regions_equal_the_single_band():
    """The mesh flavours' band dissected into regions (relax_general.hip, assign): the same LM trajectory as the single band
    (same iterations; costs equal to rounding of the different elimination order), and the dissection does trigger at 1 000
    cameras."""
    cut, r_cut = _mesh_probe({})
    band, r_band = _mesh_probe({"OCHIP_TEST_HOOKS": "no_dissect"})
    assert r_cut >= 2 and r_band == 1
    assert set(cut) == {"mesh, 4 vertices", "mesh, grid"}
    for name in cut:
        assert cut[name][0] == band[name][0], (name, cut[name], band[name])
        assert abs(cut[name][1] - band[name][1]) <= 1e-9 * abs(band[name][1]), (name, cut[name], band[name])
