"""Worker of tests/test_gpu_dp_train.py: one rank of a data-parallel training step (one generator step, then one
discriminator step, as the reference's epoch does) on a golden batch.  Launched by torch.distributed.run with the gloo
backend, every rank on cuda:0 (a one-GPU box; on a node the backend is nccl = RCCL and each rank has its GPU).
Rank 0 writes losses, gradients and the weights after optimizer.step() to the .npz named on the command line."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from util_golden import load, state_dict  # noqa: E402


def main(name, out_path):
    dist.init_process_group("gloo")
    from ddgani_amd import dist as ddist, ops
    from ddgani_amd.model.Diffusion import _NetState
    dev = torch.device("cuda:0")
    comm = ddist.Comm()
    z = load(name)
    T = int(z["T"])
    lay = ops.FieldLayout(z["widths"].tolist(), dev)
    PG = ops.GenParams(state_dict(z, "G0__"), dev)
    PD = ops.DiscParams(state_dict(z, "D0__"), dev)
    sched = torch.from_numpy(ops.schedule(T)).to(dev)
    up = lambda k, dt=None: torch.tensor(z[k] if dt is None else z[k].astype(dt)).to(dev)
    sG = _NetState(PG, ops.GEN_KEYS, 1e-4, 1e-4)
    sD = _NetState(PD, ops.DISC_KEYS, 1e-4, 0.0)
    lw = {"Res_weight": float(z["w_res"]), "G_D_weight": float(z["w_gd"])}
    x0, M, m = up("x0"), up("M"), up("m")
    lg = ddist.sharded_train_step(comm, ops, "G", PG, PD, sG, x0, M, m, up("t"), lay, sched, T, loss_weight=lw,
                                  eps=up("eps"), drops=[up(f"mask_{l}", np.uint8) for l in range(3)])
    gG = {k: v.clone() for k, v in sG.buf.g.items()}
    ld = ddist.sharded_train_step(comm, ops, "D", PG, PD, sD, x0, M, m, up("t2"), lay, sched, T, loss_scale=1.0,
                                  eps=up("eps2"), drops=[up(f"mask2_{l}", np.uint8) for l in range(3)])
    torch.cuda.synchronize()
    if comm.rank == 0:
        res = {"lg": lg.cpu().numpy(), "ld": ld.cpu().numpy(), "world": np.array(comm.world)}
        for k in ops.GEN_KEYS:
            res["G1__" + k] = PG.t[k].cpu().numpy()
            res["gG__" + k] = gG[k].cpu().numpy()
        for k in ops.DISC_KEYS:
            res["D1__" + k] = PD.t[k].cpu().numpy()
            res["gD__" + k] = sD.buf.g[k].cpu().numpy()
        np.savez(out_path, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
