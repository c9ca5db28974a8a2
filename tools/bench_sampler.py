#!/usr/bin/env python3
"""Sampler-only timing (T-step reverse diffusion + mask-merge) on a synthetic table: the part of the hot path between
attention fills.  Prints one JSON line; run under `rocprofv3 --kernel-trace --stats` for per-kernel times."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    from ddgani_amd import ops, synth
    from ddgani_amd.model import Diffusion as D_
    dev = torch.device("cuda:0")
    tab = synth.make_table(a.workload, dev, rows=a.rows)
    lay, T = tab.layout, tab.T
    P = ops.GenParams(synth.random_generator_state(lay.D, T), dev)
    sh = ops.schedule(T)
    merged = torch.empty_like(tab.x_diff)

    def step(i):
        x0 = D_.reverse_sample(P, T, tab.x_diff, tab.M, lay, seed=i, sched_host=sh)
        ops.mask_merge(tab.x_diff, tab.M, x0, lay, out=merged)
    for i in range(2):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    flops = 16.0 * lay.D * lay.D * a.rows * (T - 1)
    print(json.dumps({"what": "sampler + merge", "workload": a.workload, "rows": a.rows, "encoded_dim": lay.D, "T": T,
                      "ms_per_pass": dt * 1e3, "rows_per_s": a.rows / dt, "denoiser_gemm_flops": flops,
                      "gemm_TFLOPs_if_all_time_were_gemm": flops / dt / 1e12}))


if __name__ == "__main__":
    main()
