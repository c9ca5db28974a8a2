"""Worker of tests/test_gpu_dist_sample.py: one rank of the row-sharded T-step sampler in THROUGHPUT mode (on-chip random
numbers), gloo backend, every rank on cuda:0.  Each rank writes its rows of x0_hat to <out>.<rank>.npy."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main(rows, out_path):
    dist.init_process_group("gloo")
    from ddgani_amd import dist as ddist, ops, synth
    dev = torch.device("cuda:0")
    comm = ddist.Comm()
    tab = synth.make_table("c3", dev, rows=rows)
    P = ops.GenParams(synth.random_generator_state(tab.layout.D, tab.T), dev)
    lo, hi = ddist.shard_bounds(rows, comm.world)[comm.rank]
    out = ddist.sharded_reverse_sample(comm, ops, P, tab.T, tab.x_diff[lo:hi].contiguous(), tab.M[lo:hi].contiguous(),
                                       tab.layout, rows, seed=11, call=77)
    torch.cuda.synchronize()
    np.save(f"{out_path}.{comm.rank}.npy", np.concatenate([[lo, hi], out.cpu().numpy().ravel()]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), sys.argv[2])
