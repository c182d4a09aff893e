"""The host library's surface-mesh code (csrc/host/relax_mesh.cpp: rebuildMesh, buildMinimalMesh with an exact bucket-grid
nearest-neighbour search) against the oracle's restatement of src/surface/expand_mesh.cpp (exhaustive search).  No device."""
import numpy as np

from opencalibration_amd import host
from relax_fixtures import camera_grid


def _same(a, b):
    assert np.array_equal(a["vertices"], b["vertices"]) and np.array_equal(a["edges"], b["edges"])


def test_meshes_from_cameras_only(oracle):
    for rows, cols, seed in [(3, 4, 3), (5, 7, 1), (1, 2, 0), (9, 2, 5)]:
        _, pos, _, _ = camera_grid(rows, cols, seed=seed)
        for minimal in (True, False):
            _same(host.rebuild_mesh(pos, minimal=minimal).arrays(), oracle.rebuild_mesh(pos, minimal=minimal).arrays())
    # fewer than two cameras: no mesh
    assert len(host.rebuild_mesh(np.zeros((1, 3)), minimal=True).arrays()["vertices"]) == 0
    assert len(host.rebuild_mesh(np.zeros((1, 3)), minimal=False).arrays()["vertices"]) == 0


def test_meshes_over_a_previous_surface(oracle):
    """Heights come from the nearest vertex / cloud point of the previous surface; the border from the median height
    of the cameras above it (expand_mesh.cpp:49-121)."""
    rng = np.random.default_rng(7)
    _, pos, _, _ = camera_grid(6, 5, seed=2)
    first = host.rebuild_mesh(pos, minimal=False).arrays()
    verts = first["vertices"].copy()
    verts[:, 2] += rng.normal(0, 0.3, len(verts))
    cloud = np.column_stack([rng.uniform(verts[:, 0].min(), verts[:, 0].max(), 3000),
                             rng.uniform(verts[:, 1].min(), verts[:, 1].max(), 3000), rng.normal(-1, 0.5, 3000)])
    hp = host.Surface().set(verts, first["edges"], cloud)
    op = oracle.RxSurface().set(verts, first["edges"], cloud)
    for minimal in (True, False):
        _same(host.rebuild_mesh(pos, hp, minimal).arrays(), oracle.rebuild_mesh(pos, op, minimal).arrays())
    # a cloud alone (mesh-less previous surface) still steers the heights
    hp2 = host.Surface().set(np.zeros((0, 3)), np.zeros((0, 5), np.uint64), cloud)
    op2 = oracle.RxSurface().set(np.zeros((0, 3)), np.zeros((0, 5), np.uint64), cloud)
    _same(host.rebuild_mesh(pos, hp2, False).arrays(), oracle.rebuild_mesh(pos, op2, False).arrays())


def test_surface_round_trip():
    _, pos, _, _ = camera_grid(3, 3)
    a = host.rebuild_mesh(pos, minimal=False).arrays()
    b = host.Surface().set(a["vertices"], a["edges"], a["vertices"][:5]).arrays()
    assert np.array_equal(a["vertices"], b["vertices"]) and np.array_equal(a["edges"], b["edges"])
    assert np.array_equal(b["cloud"], a["vertices"][:5])
