"""GPU: the data-parallel training step (ddg_train_gen_grads_dp / ddg_train_disc_grads_dp +
ddgani_amd.dist.sharded_train_step): the reference's B = 128 minibatch sharded by rows over 2 and 3 ranks with
synchronised BatchNorm and one gradient all-reduce.  Ranks are real processes (torch.distributed.run, gloo, all on
cuda:0: the box has one GPU; the production backend is nccl = RCCL).  Asserted against the REFERENCE's own step
(tests/golden/train_*.npz): losses 2e-6, gradients 1e-4 relative, weights after optimizer.step() 2e-7 -- the same
bars as the single-process step (tests/test_gpu_train.py) -- and against the single-process step of this library."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from util_golden import load, state_dict

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def _single(z, dev):
    from ddgani_amd import ops
    from ddgani_amd.model.Diffusion import _NetState
    T = int(z["T"])
    lay = ops.FieldLayout(z["widths"].tolist(), dev)
    PG, PD = ops.GenParams(state_dict(z, "G0__"), dev), ops.DiscParams(state_dict(z, "D0__"), dev)
    sched = torch.from_numpy(ops.schedule(T)).to(dev)
    up = lambda k, dt=None: torch.tensor(z[k] if dt is None else z[k].astype(dt)).to(dev)
    sG, sD = _NetState(PG, ops.GEN_KEYS, 1e-4, 1e-4), _NetState(PD, ops.DISC_KEYS, 1e-4, 0.0)
    ops.train_gen_grads(PG, PD, sG.buf, up("x0"), up("M"), up("m"), up("t"), lay, sched, T, float(z["w_res"]),
                        float(z["w_gd"]), eps=up("eps"), drops=[up(f"mask_{l}", np.uint8) for l in range(3)])
    sG.step()
    ops.train_disc_grads(PG, PD, sD.buf, up("x0"), up("M"), up("m"), up("t2"), lay, sched, T, loss_scale=1.0,
                         eps=up("eps2"), drops=[up(f"mask2_{l}", np.uint8) for l in range(3)])
    sD.step()
    return ({k: PG.t[k].cpu().numpy() for k in ops.GEN_KEYS}, {k: PD.t[k].cpu().numpy() for k in ops.DISC_KEYS})


@pytest.mark.parametrize("name,world", [("train_mixed", 2), ("train_num", 2), ("train_mixed", 3)])
def test_data_parallel_step_matches_reference_and_single_process(tmp_path, name, world):
    from ddgani_amd import ops
    out = str(tmp_path / "dp.npz")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "dp_train_worker.py"), name, out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = np.load(out)
    assert int(d["world"]) == world
    z = load(name)
    # against the reference's step
    assert abs(d["lg"][0] - float(z["res_loss"])) < 2e-6 and abs(d["lg"][1] - float(z["gd_loss"])) < 2e-6, d["lg"]
    assert abs(d["ld"][0] - float(z["d_loss"])) < 2e-6, d["ld"]
    noise = ("linears.0.bias", "linears.4.bias", "linears.8.bias")      # true gradient 0 (bias before a BatchNorm)
    for k in ops.GEN_KEYS:
        kk = k.replace(".", "__")
        if k not in noise:
            assert _rel(d["gG__" + k], z["gG__" + kk]) < 1e-4, (k, _rel(d["gG__" + k], z["gG__" + kk]))
            g = z["gG__" + kk] + 1e-4 * z["G0__" + kk]
            stable = np.abs(g) > 1e-6 * max(np.abs(g).max(), 1e-30) + 1e-7
            assert np.abs(d["G1__" + k] - z["G1__" + kk])[stable].max() < 2e-7, k
        assert np.abs(d["G1__" + k] - z["G1__" + kk]).max() < 2.1e-4, k
    for k in ops.DISC_KEYS:
        kk = k.replace(".", "__")
        # the reference's D step ran on ITS updated generator; ours on the data-parallel one: same to 2e-7, so the
        # discriminator gradients agree to the single-process bar as well
        assert _rel(d["gD__" + k], z["gD__" + kk]) < 1e-4, (k, _rel(d["gD__" + k], z["gD__" + kk]))
        g = z["gD__" + kk]
        stable = np.abs(g) > 1e-6 * max(np.abs(g).max(), 1e-30) + 1e-9
        assert np.abs(d["D1__" + k] - z["D1__" + kk])[stable].max() < 2e-7, k
    # against this library's single-process step on the same batch: same weights wherever Adam's first step is
    # not sign-ambiguous
    G1, D1 = _single(z, torch.device("cuda:0"))
    for k in ops.GEN_KEYS:
        if k in noise:
            continue
        kk = k.replace(".", "__")
        g = z["gG__" + kk] + 1e-4 * z["G0__" + kk]
        stable = np.abs(g) > 1e-6 * max(np.abs(g).max(), 1e-30) + 1e-7
        assert np.abs(d["G1__" + k] - G1[k])[stable].max() < 2e-7, k
    for k in ops.DISC_KEYS:
        g = z["gD__" + k.replace(".", "__")]
        stable = np.abs(g) > 1e-6 * max(np.abs(g).max(), 1e-30) + 1e-9
        assert np.abs(d["D1__" + k] - D1[k])[stable].max() < 2e-7, k
