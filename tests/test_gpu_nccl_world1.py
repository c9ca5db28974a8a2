"""GPU: the `nccl` (= RCCL) branch of ddgani_amd.dist executes -- with one rank, which is all a one-GPU box can host (two
RCCL ranks cannot share a device).  See tests/dist_nccl_worker.py for what is checked.  Multi-rank behaviour of the same
code is covered over gloo (tests/test_gpu_dist_sample.py, tests/test_gpu_dp_train.py, tests/test_dist_gloo.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_rccl_collectives_of_the_sharded_path_with_one_rank():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0",
               LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "dist_nccl_worker.py"), "20000"], env=env, capture_output=True,
                       text=True, timeout=240)
    assert r.returncode == 0 and "NCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
