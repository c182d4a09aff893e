"""Eight ranks - the node BASELINE's scaling curve is quoted on - on the one GPU of the test box (gloo transport, the ranks
share cuda:0): one survey from pixels over the ranks, its relaxes sharded, the clustered stage's groups dealt over the ranks
(tests/sharded_world8_worker.py), and `bench.py --gpus 8` end to end.  Two RCCL ranks have never run (1-GPU boxes); this
keeps everything around the transport - block arithmetic, exchanges, launch, the line's format - exercised at N = 8."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eight_ranks_survey_relaxes_and_groups():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "tests", "sharded_world8_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="2", OCHIP_HOST_THREADS="2")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "WORLD8 OK" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_bench_gpus_8_prints_one_line_with_both_curves():
    """`bench.py --gpus 8 --config C1` the way the scaling run calls it (no launcher around it), on gloo: ONE JSON line with the
    weak value and the strong section (one survey over all ranks) - nothing about N = 8 may fail for a trivial reason."""
    env = dict(os.environ, OCHIP_BENCH_BACKEND="gloo", OCHIP_HOST_THREADS="2", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "C1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-3000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["value"] > 0
    ss = line["strong_scaling"]
    assert "error" not in ss, ss
    assert ss["images_per_s"] > 0 and len(ss["seconds_per_step_per_rank"]) == 8
    assert line["images_per_s_weak_one_survey_per_gpu"] == line["value"]
    assert line["images_per_s_strong_one_survey_over_all_gpus"] == ss["images_per_s"]
    assert line["collective_preflight_s"] is not None
