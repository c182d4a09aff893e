"""RelaxStage::init's partition on the host (csrc/host/relax_stage.cpp: k-means restated, spectral embedding by a Lanczos
eigensolver in place of Spectra) against the oracle (oracle/relax_cluster.cpp: the same k-means - pinned bit for bit against
the reference's own KMeans.hpp - and a dense Jacobi eigensolver): two independent eigensolvers must lead to the same
groups.  No device."""
import numpy as np
import pytest

from opencalibration_amd import host
from relax_fixtures import (MODEL_600, camera_grid, grid_5x5, host_graph_from_edges, host_paths, rx_graph_from_edges)


def _groups_as_sets(assign):
    out = {}
    for node, g in enumerate(assign):
        out.setdefault(int(g), set()).add(node)
    return sorted((sorted(v) for k, v in out.items() if k >= 0), key=lambda v: (-len(v), v))


@pytest.mark.parametrize("rows,cols", [(10, 12), (7, 16), (12, 18)])  # (a square grid has a degenerate spectrum: any rotation of its x- and y-modes is an eigenbasis, the split is then the eigensolver's pick - Spectra's as much as anyone's)
def test_partition_matches_oracle(oracle, rows, cols):
    ori, pos, edges, model = camera_grid(rows, cols, pts_per_side=2 * max(rows, cols))
    n = rows * cols
    rx, _ = rx_graph_from_edges(oracle, pos, ori, model, edges, paths=host_paths(n))
    k_exp, assign_exp, depth = oracle.relax_stage_groups(rx)
    assert k_exp == n // 50 and depth == 0
    g = host_graph_from_edges(host, pos, ori, model, edges)
    k_got, assign_got = g.relax_partition(k_exp)
    assert k_got == k_exp
    assert _groups_as_sets(assign_got) == _groups_as_sets(assign_exp)
    sizes = [len(s) for s in _groups_as_sets(assign_got)]
    assert sum(sizes) == n and min(sizes) > 10      # a partition into compact groups, none degenerate
    # largest group first (relax_stage.cpp:100)
    counts = np.bincount(assign_got)
    assert all(counts[i] >= counts[i + 1] for i in range(len(counts) - 1))
    g.close()


def test_two_disconnected_surveys_split_first(oracle):
    """Connected components of the link graph get their own share of the clusters (spectral_cluster.hpp:163-214)."""
    o1, p1, e1, model = camera_grid(6, 10, seed=1, pts_per_side=24)
    o2, p2, e2, _ = camera_grid(5, 10, seed=2, pts_per_side=24)
    p2 = p2 + np.array([200.0, 0, 0])
    n1 = len(p1)
    for e in e2:
        e["src"] += n1
        e["dst"] += n1
    pos, ori, edges = np.concatenate([p1, p2]), np.concatenate([o1, o2]), e1 + e2
    rx, _ = rx_graph_from_edges(oracle, pos, ori, model, edges)
    k_exp, assign_exp, _ = oracle.relax_stage_groups(rx)
    g = host_graph_from_edges(host, pos, ori, model, edges)
    k_got, assign_got = g.relax_partition(k_exp)
    assert k_got == k_exp == 2
    assert _groups_as_sets(assign_got) == _groups_as_sets(assign_exp) == [list(range(n1)), list(range(n1, len(pos)))]
    g.close()
