#!/usr/bin/env python3
"""Microbenchmark of the head + reverse-step kernels at 1M rows: on-chip noise vs explicit z (one more 4-byte stream, no
Philox / Box-Muller), c3's mixed layout vs an all-numeric one of the same width -- tells VALU-bound from HBM-bound.
    python tools/bench_head.py [--rows 1000000]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from ddgani_amd import ops, synth
    dev = torch.device("cuda:0")
    tab = synth.make_table("c3", dev, rows=a.rows)
    lay, T = tab.layout, tab.T
    D = lay.D
    num = ops.FieldLayout([1] * D)
    y = torch.randn(a.rows, D, device=dev)
    xt = torch.randn(a.rows, D, device=dev)
    z = torch.randn(a.rows, D, device=dev)
    out = torch.empty_like(xt)
    sh = ops.schedule(T)

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    res = {}
    for nm, L in (("c3_mixed", lay), ("all_numeric", num)):
        res[f"head_reverse/{nm}/chip_noise"] = timed(lambda: ops.head_reverse_step(y, xt, tab.x_diff, tab.M, 4, T, L, seed=1, stream_id=2, out=out, sched_host=sh))
        res[f"head_reverse/{nm}/explicit_z"] = timed(lambda: ops.head_reverse_step(y, xt, tab.x_diff, tab.M, 4, T, L, z=z, out=out, sched_host=sh))
        res[f"head/{nm}"] = timed(lambda: ops.head(y, L, out=out))
        res[f"mask_merge/{nm}"] = timed(lambda: ops.mask_merge(tab.x_diff, tab.M, y, L, out=out))
    res["reverse_step/chip_noise"] = timed(lambda: ops.reverse_step(xt, tab.x_diff, tab.M, y, 4, T, seed=1, stream_id=2, out=out, sched_host=sh))
    res["reverse_step/explicit_z"] = timed(lambda: ops.reverse_step(xt, tab.x_diff, tab.M, y, 4, T, z=z, out=out, sched_host=sh))
    res["forward_noise/chip"] = timed(lambda: ops.forward_noise(tab.x_diff, T, T, seed=1, stream_id=3, out=out))
    n = a.rows * D
    gbs = {k: (n * (24 if "explicit_z" in k and "head_reverse" in k else 20 if "head_reverse" in k else 24 if k == "reverse_step/explicit_z" else
                    20 if k.startswith("reverse_step") else 16 if k.startswith("mask_merge") else 8) / (v * 1e-3) / 1e9) for k, v in res.items()}
    print(json.dumps({"rows": a.rows, "D": D, "ms": res, "GBps_algorithmic": gbs}, indent=1))


if __name__ == "__main__":
    main()
