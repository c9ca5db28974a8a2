"""Worker of tests/test_gpu_nccl_world1.py: the row-sharded path over the `nccl` (= RCCL) backend with ONE rank -- the only
RCCL configuration a one-GPU box can run.  With `Comm.on` forced, every collective of ddgani_amd.dist really goes through
RCCL on device memory (all-gather of fp32 key rows and uint8 masks, all-reduce of fp64 BatchNorm sums and of the fp32
gradient buffer, fp64 / int64 reductions of bench.py's timing), and the results must equal the single-process path."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main(rows):
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    from ddgani_amd import dist as ddist, ops, synth
    assert dist.get_backend() == "nccl"
    comm = ddist.Comm()
    assert not comm.on and not comm.host_staged        # one rank: off by default ...
    comm.on = True                                     # ... forced on: every collective below is an RCCL call
    tab = synth.make_table("c3", dev, rows=rows)
    lay = tab.layout
    # raw collectives with the dtypes the path uses
    g = comm.all_gather_rows(tab.m8, [rows])
    assert g.dtype == torch.uint8 and torch.equal(g, tab.m8)
    g = comm.all_gather_rows(tab.X_attn, [rows])
    assert torch.equal(g, tab.X_attn)
    s = torch.arange(8, dtype=torch.float64, device=dev)
    assert torch.equal(comm.all_reduce_sum(s.clone()), s)
    tt = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    one = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    assert float(tt.item()) == 1.5 and int(one.item()) == 1
    dist.barrier()
    # attention fill: sharded entry (keys through the all-gather) == plain call, bit for bit
    a = ddist.sharded_attn_fill(comm, ops, tab.X_attn, tab.m8, tab.W, lay, tab.k, rows)
    b = ops.attn_topk_fill(tab.X_attn, tab.m8, tab.X_attn, tab.m8, tab.W, lay, tab.k)
    assert torch.equal(a, b)
    # sampler with the BatchNorm sums through RCCL == the same call without collectives
    P = ops.GenParams(synth.random_generator_state(lay.D, tab.T), dev)
    x1 = ddist.sharded_reverse_sample(comm, ops, P, tab.T, tab.x_diff, tab.M, lay, rows, seed=11, call=77)
    x0 = ddist.sharded_reverse_sample(ddist.Comm(), ops, P, tab.T, tab.x_diff, tab.M, lay, rows, seed=11, call=77)
    torch.cuda.synchronize()
    d = float((x1 - x0).abs().max())
    assert np.isfinite(x1.cpu().numpy()).all() and d < 2e-5, d
    print("NCCL_WORLD1_OK", d, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]))
