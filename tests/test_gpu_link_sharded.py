"""Strong scaling of ONE survey's link stage on the device path (BASELINE config C4): directed pairs
sharded by source-image block over 2 ranks that share the box's GPU, every rank linking its block with LinkStage on
libochip, edges all-gathered; the merged graph equals the single-process graph bit for bit (ids included) and relaxes to
the same orientations."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_link_one_survey():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "tests", "sharded_link_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="8")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "SHARDED_LINK OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
