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


def test_rccl_branch_of_the_exchange_runs_on_the_device_buffers():
    """One rank on the "nccl" backend (RCCL): every exchange of the solve is an all_gather_into_tensor on the solver's
    own device arrays through the CUDA array interface - the branch a multi-GPU node takes.  A 1-GPU box cannot hold two
    RCCL ranks (one device per rank), so this covers the transport and the buffer hand-over, the 2-rank test above the
    sharding arithmetic."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(root, "tests", "sharded_relax_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="8", OCHIP_TEST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "SHARDED_RELAX OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_native_rccl_exchange_inside_the_library():
    """libochip's own communicator (ochip_rccl_comm_create, ochip_rccl_relax_exchange): every evaluation of the solve
    enqueues its ncclAllGather group on the solver's stream; no Python callback, no host wait.  One rank, for the same
    reason as above; the result must equal the unsharded solve to the bit and the communicator must have been used."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29545", os.path.join(root, "tests", "sharded_relax_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="8", OCHIP_TEST_BACKEND="gloo", OCHIP_TEST_EXCHANGE="rccl",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "SHARDED_RELAX OK" in out.stdout and "RCCL_EXCHANGES" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
