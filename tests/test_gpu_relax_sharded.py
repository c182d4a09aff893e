"""The one exchange step of the path: a single-group relax with its residual blocks sharded over ranks
(ochip_relax_set_shard).  Two processes (torch.distributed, gloo rendezvous on 127.0.0.1) share the box's GPU;
each evaluates half of the camera pairs, the per-pair records are all-gathered, and both must reproduce the
unsharded result bit for bit (the assembly after the exchange is the same deterministic code on the same records)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_reproduce_the_unsharded_relax():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "tests", "sharded_relax_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="8")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "SHARDED_RELAX OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
