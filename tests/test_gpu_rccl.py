"""GPU: the slab pipeline's collectives on a 1-rank RCCL ("nccl") process group -- all_to_all_single, scatter, gather,
all_gather_into_tensor and broadcast with the tensors and views the pipeline really passes (gloo accepts things RCCL does not).
Real multi-GPU runs need more than this box's one GPU."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pipeline_collectives_on_a_one_rank_rccl_group():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "nccl_world1_check.py")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "exchange=tiles: RCCL collectives ok" in r.stdout and "exchange=all_gather: RCCL collectives ok" in r.stdout
