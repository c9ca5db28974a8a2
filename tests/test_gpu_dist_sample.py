"""GPU: throughput-mode results of the row-sharded sampler do not depend on the world size.  On-chip random numbers
are keyed by (seed, stream, local element index); ddgani_amd.dist cuts the table into 8 fixed virtual shards, one launch
and one stream number each, and gives every rank 8 / world consecutive ones -- so 1, 2 and 4 ranks draw the same numbers
for the same rows.  Ranks are real processes (torch.distributed.run, gloo, all on cuda:0: the box has one GPU).
What may differ is the fp64 summation order of the BatchNorm statistics (atomics / all-reduce): 2e-5 on x0_hat."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROWS = 30_001          # not a multiple of 8: uneven virtual shards


def _run(world, tmp_path):
    out = str(tmp_path / f"w{world}")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "dist_sample_worker.py"), str(ROWS), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    parts, nxt = [], 0
    for rank in range(world):
        a = np.load(f"{out}.{rank}.npy")
        lo, hi = int(a[0]), int(a[1])
        assert lo == nxt
        nxt = hi
        parts.append(a[2:].reshape(hi - lo, -1))
    assert nxt == ROWS
    return np.concatenate(parts, 0)


def test_sampler_results_do_not_depend_on_world_size(tmp_path):
    from ddgani_amd import dist as ddist, ops, synth
    dev = torch.device("cuda:0")
    tab = synth.make_table("c3", dev, rows=ROWS)
    P = ops.GenParams(synth.random_generator_state(tab.layout.D, tab.T), dev)
    one = ddist.sharded_reverse_sample(ddist.Comm(), ops, P, tab.T, tab.x_diff, tab.M, tab.layout, ROWS, seed=11,
                                       call=77).cpu().numpy()
    assert np.isfinite(one).all()
    for world in (2, 4):
        many = _run(world, tmp_path)
        assert many.shape == one.shape
        d = np.abs(many - one).max()
        assert d < 2e-5, (world, d)
    # a different call number draws different numbers (the check above is not vacuous)
    other = ddist.sharded_reverse_sample(ddist.Comm(), ops, P, tab.T, tab.x_diff, tab.M, tab.layout, ROWS, seed=11,
                                         call=78).cpu().numpy()
    assert np.abs(other - one).max() > 1e-3
