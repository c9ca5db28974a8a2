#!/usr/bin/env python3
"""Training-step throughput ("plus, separately, training steps/s"): generator +
discriminator minibatch steps (B = 128, model/Diffusion.py:287-311 / :356-372) on a synthetic table,
on-chip random numbers.  Prints one JSON line.  Not the headline metric (bench.py is)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def measure(tab, dev, comm, epochs, warm_epochs=3):
    """seconds for `epochs` epochs of T generator + T discriminator minibatch steps (B = 128, Adam after each) on the
    table `tab`, on-chip random numbers, the throughput-mode loop of model/Diffusion.py (graph replay on one rank)"""
    from ddgani_amd import ops
    from ddgani_amd.model import Diffusion as D_, Discriminator_model as DM
    lay, T = tab.layout, tab.T

    class F:
        def __init__(s, w): s.w = w; s.data_type = "Categorical Data" if w > 1 else "Numerical Data"
        def dim(s): return s.w
    torch.manual_seed(3407)
    G = D_.Generator_x0(lay.D, 2 * lay.D, lay.D, T + 1, [F(w) for w in lay.widths]).to(dev)
    Dn = DM.Discriminator_noise_x(2 * lay.D, lay.D // 2, lay.C, T + 1).to(dev)
    optG = D_._NetState(G.device_params(refresh=True), ops.GEN_KEYS, 1e-4, 1e-4)
    optD = D_._NetState(Dn.device_params(refresh=True), ops.DISC_KEYS, 1e-4, 0.0)
    DS = D_.Diffusion_setting(T, lay.D, dev)
    m_data = tab.m8.float()
    lw = {"Res_weight": 100, "G_D_weight": 1, "D_weight": 0.2, "L_weight": 0.1}
    rnd = D_.ChipRandom(0)

    def epoch(e):
        D_.train_one_epoch_Generator(G, Dn, DS, tab.x_diff, optG, None, 0.0, e, epochs, T, dev, None, tab.M, m_data, lw,
                                     [], None, None, rnd=rnd, comm=comm)
        D_.train_one_epoch_discriminator_noise_x(G, Dn, DS, tab.x_diff, optD, None, 0.0, e, epochs, T, dev, None, tab.M,
                                                 m_data, lw, rnd=rnd, comm=comm)
    try:
        for e in range(1, warm_epochs + 1):
            epoch(e)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for e in range(1, epochs + 1):
            epoch(e)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    finally:
        gs = getattr(rnd, "_graphs", None)
        if gs is not None:
            gs.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--rows", type=int, default=200_000)
    ap.add_argument("--epochs", type=int, default=50)
    ap.add_argument("--gpus", type=int, default=1, help="data-parallel ranks (sync-BatchNorm + gradient all-reduce)")
    ap.add_argument("--backend", default="nccl", help="nccl = RCCL; gloo with --one-device rehearses on a one-GPU box")
    ap.add_argument("--one-device", action="store_true")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:       # start the ranks as a child, before touching the GPU
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                  f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
                                  os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if a.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
    from ddgani_amd import synth
    from ddgani_amd import dist as ddist
    comm = ddist.Comm()
    tab = synth.make_table(a.workload, dev, rows=a.rows)
    lay, T = tab.layout, tab.T
    dt = measure(tab, dev, comm, a.epochs)
    steps = a.epochs * 2 * T
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if comm.rank == 0:
      print(json.dumps({"metric": "training steps/s (generator + discriminator minibatch steps, B=128)",
                      "value": steps / dt, "unit": "steps/s", "ms_per_step": dt / steps * 1e3,
                      "config": {"workload": a.workload, "rows": a.rows, "encoded_dim": lay.D, "cols": lay.C, "T": T,
                                 "epochs": a.epochs, "rng": "chip",
                                 "parallelism": f"data-parallel x{world} (B=128 sharded by rows, sync-BN, one gradient "
                                                f"all-reduce per step, backend {a.backend})" if world > 1 else "single GPU"},
                      "n_gpus": world, "dtype": "f32", "data": "synthetic"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
