#!/usr/bin/env python3
"""A BASELINE workload at its FULL size on one GPU (c4: 10M rows x 128 columns, MCAR 0.5; c5: 2M x 256, Region), one step
of the hot path timed per phase, and a parity check of a query-row shard that is NOT a self-comparison:

 * the shard is taken where the table HAS missing cells (c5's Region rectangle starts past row 0);
 * the screened fill of the whole table is compared with the fp32 path run for the shard's rows as queries over ALL keys,
   and EVERY cell that differs by more than 1e-5 is adjudicated in fp64 on the device: the pair's logits are recomputed
   in fp64 over all valid keys and the cell is accepted only if its k-th and (k+1)-th largest logits are closer than
   1e-6 (the north star's tie exemption) -- the gap and the fp64 fill are recorded;
 * `--spot` random (row, column) pairs of the shard are held to the fp64 closed form outright (k-th logit 1e-6, fill 1e-5
   plus the weight of the keys within 1e-6 of it).

    python tools/full_size_check.py --workload c4 --shard 1024                     # whole step + shard (k as the workload names it)
    python tools/full_size_check.py --workload c4 --topk 0.05 --shard 16384 --no-full   # the reference's top_k on a query shard

With --no-full only the shard runs (the reference's top_k = 0.05 at 10M rows is two full similarity passes: hours of a
single GPU for the whole table, an 8-GPU node shards the queries); its time is reported with the per-row extrapolation.
Writes one JSON line."""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def fp64_pair(tab, row, c, k):
    """fp64 closed form of one (row, column) pair on the device -> (kth, gap to the (k+1)-th, fill [d_c], near-tie slack)"""
    off = tab.layout.offsets
    keys = torch.nonzero(tab.m8[:, c] == 1).flatten()
    keys = keys[keys != row]
    q = tab.W[c].double() * tab.X_attn[row].double()
    s = torch.zeros(keys.numel(), dtype=torch.float64, device=keys.device)
    step = 1 << 20                                        # chunks: X64 of the whole table would not be needed at once
    for i in range(0, keys.numel(), step):
        s[i:i + step] = tab.X_attn[keys[i:i + step]].double() @ q
    kk = min(k, keys.numel())
    top = torch.topk(s, min(kk + 1, keys.numel())).values
    kth = top[kk - 1]
    gap = float(top[kk - 1] - top[kk]) if keys.numel() > kk else float("inf")
    keep = s >= kth
    w = torch.exp(s - s.max())
    xv = tab.X_attn[keys][:, off[c]:off[c + 1]].double()
    Z = w[keep].sum()
    ref = (w[keep, None] * xv[keep]).sum(0) / Z
    band = (s - kth).abs() <= 1e-6
    slack = (w[band, None] * (xv[band] - ref).abs()).sum(0) / Z + 1e-5
    return float(kth), gap, ref.cpu().numpy(), slack.cpu().numpy(), int(keys.numel())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--shard", type=int, default=1024)
    ap.add_argument("--topk", type=float, default=0.0, help="< 1: the reference's fraction of N; >= 1: a count; 0: the workload's")
    ap.add_argument("--no-full", action="store_true", help="only the query shard (timed), not the whole-table step")
    ap.add_argument("--spot", type=int, default=8, help="random pairs of the shard against the fp64 closed form")
    ap.add_argument("--max-adjudicate", type=int, default=64)
    a = ap.parse_args()
    from ddgani_amd import ops, synth
    from ddgani_amd.model import Diffusion as D_
    dev = torch.device("cuda:0")
    wl = synth.WORKLOADS[a.workload]
    N = a.rows or wl["rows"]
    t0 = time.perf_counter()
    tab = synth.make_table(a.workload, dev, rows=N)
    torch.cuda.synchronize()
    print(f"table built in {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
    lay, k, T = tab.layout, tab.k, tab.T
    if a.topk > 0:
        k = int(a.topk) if a.topk >= 1 else max(1, int(N * a.topk))
    done = threading.Event()
    t_start = time.perf_counter()

    def ticker():      # the select phase enqueues hundreds of launches; keep the run visibly alive
        while not done.wait(30):
            print(f"... {time.perf_counter() - t_start:.0f} s", file=sys.stderr, flush=True)
    threading.Thread(target=ticker, daemon=True).start()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    res = {"workload": f"{a.workload}: {wl['desc']}" + (f" [rows {N}]" if a.rows else ""), "rows": N, "cols": lay.C,
           "encoded_dim": lay.D, "top_k": k, "T": T, "n_gpus": 1}
    # the shard: R consecutive rows starting at the first row that has a missing cell
    has_missing = (tab.m8 == 0).any(1)
    first = int(torch.nonzero(has_missing).flatten()[0].item())
    R = min(a.shard, N - first)
    lo, hi = first, first + R
    out_full = None
    if not a.no_full:
        e = [ev() for _ in range(5)]
        out_code = torch.empty_like(tab.X_attn)
        plan = ops.AttnPlan(tab.X_attn, tab.m8, tab.X_attn, tab.m8, tab.W, lay, k, mode=ops.ATTN_BF16_FILTER)
        print(f"attention workspace {plan.ws.numel() / 2**30:.1f} GiB", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        e[0].record(); plan.prepare()
        e[1].record(); plan.select(out=out_code)
        e[2].record(); plan.fill()
        e[3].record()
        x0_hat = D_.reverse_sample(P := ops.GenParams(synth.random_generator_state(lay.D, T), dev), T, tab.x_diff, tab.M, lay, seed=0)
        merged = ops.mask_merge(tab.x_diff, tab.M, x0_hat, lay)
        e[4].record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ph = {"prepare": e[0].elapsed_time(e[1]), "select": e[1].elapsed_time(e[2]), "fill": e[2].elapsed_time(e[3]),
              "sampler+merge": e[3].elapsed_time(e[4])}
        nobs = tab.m8.sum(0).double()
        flops = 2.0 * lay.D * float(((N - nobs) * nobs).sum().item())
        res.update({"seconds_per_step": dt, "rows_per_s": N / dt, "phases_ms": ph, "select_flops": flops,
                    "select_PFLOPs": flops / (ph["select"] * 1e-3) / 1e15,
                    "frac_of_16bit_mfma_peak": flops / (ph["select"] * 1e-3) / 2.5e15,
                    "fallback_pairs": ops.attn_fallback_pairs(N, N, lay, k, True, dev)})
        out_full = out_code[lo:hi].clone()
        del plan, x0_hat, merged, out_code, P
        ops._ws_cache.clear()
        torch.cuda.empty_cache()
    Xq, mq = tab.X_attn[lo:hi].contiguous(), tab.m8[lo:hi].contiguous()
    miss = torch.repeat_interleave(mq == 0, torch.tensor(lay.widths, device=dev), dim=1)
    shard = {"query_rows": R, "first_row": lo, "keys": N, "missing_cells": int(miss.sum().item())}
    assert shard["missing_cells"] > 0
    # the shard through the default (screened / window) path, timed
    e0, e1 = ev(), ev()
    ops.attn_topk_fill(Xq[:256].contiguous(), mq[:256].contiguous(), tab.X_attn, tab.m8, tab.W, lay, k, q_row_offset=lo)   # warm-up
    e0.record()
    got, tau, nval = ops.attn_topk_fill(Xq, mq, tab.X_attn, tab.m8, tab.W, lay, k, q_row_offset=lo, debug=True,
                                        mode=ops.ATTN_BF16_FILTER)
    e1.record()
    torch.cuda.synchronize()
    shard["seconds_default_path"] = e0.elapsed_time(e1) * 1e-3
    shard["fallback_pairs"] = ops.attn_fallback_pairs(R, N, lay, k, False, dev)
    shard["pairs"] = int((mq == 0).sum().item())
    if a.no_full:
        res["shard_seconds"] = shard["seconds_default_path"]
        res["extrapolated_seconds_per_step_attention_only"] = shard["seconds_default_path"] * N / R
        res["note"] = ("attention fill of the shard's rows as queries over ALL keys; an N-row table costs N / shard times as "
                       "much on one GPU (queries are independent), an 8-GPU node an eighth of that")
    else:
        shard["full_run_equals_shard_run_beyond_1e-5"] = int(((out_full - got).abs()[miss] > 1e-5).sum().item())
    large_k = k > ops.MAX_TOPK_REG
    if not large_k:
        # fp32 path of the same shard; every differing cell is adjudicated in fp64
        ref, tau32, _ = ops.attn_topk_fill(Xq, mq, tab.X_attn, tab.m8, tab.W, lay, k, q_row_offset=lo, debug=True,
                                           mode=ops.ATTN_F32)
        torch.cuda.synchronize()
        src = out_full if out_full is not None else got
        diff = (src - ref).abs()
        shard["max_abs_diff_vs_fp32_path"] = float(diff[miss].max().item())
        shard["observed_cells_untouched"] = bool(torch.equal(src[~miss], Xq[~miss]))
        off = lay.offsets
        cell = torch.stack([diff[:, off[c]:off[c + 1]].max(1).values for c in range(lay.C)], 1)
        bad = torch.nonzero((cell > 1e-5) & (mq == 0))
        shard["cells_beyond_1e-5"] = int(bad.shape[0])
        adj = []
        for j, c in bad[:a.max_adjudicate].tolist():
            kth, gap, fill64, slack, nkeys = fp64_pair(tab, lo + j, c, k)
            adj.append({"row": lo + j, "col": c, "diff": float(cell[j, c]), "kth_to_next_gap_fp64": gap,
                        "screened_err_vs_fp64": float(np.abs(src[j, off[c]:off[c + 1]].double().cpu().numpy() - fill64).max()),
                        "fp32_path_err_vs_fp64": float(np.abs(ref[j, off[c]:off[c + 1]].double().cpu().numpy() - fill64).max())})
            assert gap <= 1e-6, f"cell ({lo + j}, {c}) differs by {float(cell[j, c]):.3g} without a near-tie (gap {gap:.3g})"
        shard["adjudicated_in_fp64"] = adj
    # spot pairs against the fp64 closed form
    rng = np.random.default_rng(3)
    mq_h = mq.cpu().numpy()
    rows_m = np.where((mq_h == 0).any(1))[0]
    spots, off = [], lay.offsets
    for j in rng.choice(rows_m, min(a.spot, len(rows_m)), replace=False):
        c = int(rng.choice(np.where(mq_h[j] == 0)[0]))
        kth, gap, fill64, slack, nkeys = fp64_pair(tab, lo + int(j), c, k)
        err = np.abs(got[j, off[c]:off[c + 1]].double().cpu().numpy() - fill64)
        spots.append({"row": lo + int(j), "col": c, "valid_keys": nkeys, "tau_err": abs(float(tau[j, c]) - kth),
                      "fill_err": float(err.max()), "slack": float(slack.max())})
        assert abs(float(tau[j, c]) - kth) <= 1e-6 and np.all(err <= slack), spots[-1]
    shard["fp64_spot_pairs"] = spots
    res["parity_shard"] = shard
    done.set()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
