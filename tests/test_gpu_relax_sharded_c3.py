"""BASELINE config C4 at its size: the relax of a 1 000-camera survey over 2 ranks (gloo, sharing the box's GPU) - the
single-group plane solve, the single-group ground-mesh solve (the general engine sharded) and the clustered stage with its
20 groups dealt over the ranks - each bit-identical to the one-process result on every rank."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(grid, feats, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "sharded_relax_c3_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="8", SHARD_TEST_GRID=grid, SHARD_TEST_FEATS=str(feats))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "SHARDED_RELAX_C3 OK" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_relax_of_a_1000_camera_survey_over_two_ranks():
    _run("25x40", 1024, 29561)


def test_relax_over_two_ranks_small():
    _run("10x12", 512, 29563)     # 120 cameras, 2 groups: one per rank
