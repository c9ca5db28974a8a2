#!/usr/bin/env python3
"""Benchmark of the imputation hot path: top-k row-attention fill + T-step reverse diffusion +
mask-merge over a synthetic table (BASELINE.json metric "imputed rows/sec").

    python bench.py --gpus N --steps K --warmup W [--workload c3|c4|c5] [--rows R]

One step = one pass of the hot path over the whole table (inputs resident in HBM).  For N > 1 the
driver launches one rank per GPU with torch.distributed.run; rows are sharded, key rows are
all-gathered per step (RCCL) and BatchNorm sums all-reduced: strong scaling on a fixed table.
Rank 0 prints ONE JSON line (contract in the task statement), with `roofline` for the dominant
kernel (the similarity / selection pass of the attention fill) and `cpu_baseline` (the oracle =
CPU port issuing the reference's ATen op sequence, timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide, "Peak BF16/FP16 MFMA", dense (the screen runs on f16 since round 2: same peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--rows", type=int, default=0, help="override the workload's row count (0 = as named)")
    ap.add_argument("--attn-mode", default="bf16filter", choices=["f32", "bf16filter"],
                    help="similarity arithmetic of the attention fill (include/ddgani.h: DDG_ATTN_*)")
    ap.add_argument("--topk", type=float, default=0.0,
                    help="override the workload's top_k: a value < 1 is the reference's fraction of N (param.json uses "
                         "0.05), a value >= 1 an absolute count; 0 = as the workload names it")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the "
                                                     "multi-rank path on a one-GPU box together with --one-device)")
    ap.add_argument("--one-device", action="store_true", help="every rank uses cuda:0 (rehearsal on a one-GPU box)")
    ap.add_argument("--even-shards", action="store_true", help="attention queries sharded by even row counts instead of equal "
                                                               "work (A/B: Region missingness leaves most ranks idle)")
    ap.add_argument("--cpu-rows", type=int, default=8192, help="sample rows for the CPU attention-fill timing")
    return ap.parse_args()


def cpu_baseline(tab_cfg, lay_widths, k_frac_rows, T, D, N_full, sample_rows):
    """Oracle (port of the reference's torch-CPU op sequence) on a bounded sample; rank 0 only.
    Attention fill is O(N^2): timed at N_s x N_s and extrapolated per row (cost/row grows linearly in
    N); the sampler is O(N) and timed on 100k rows."""
    import numpy as np
    from oracle import ddgani_oracle as orc
    # all host cores.  On the GPU pool os.cpu_count() reports the whole machine while a one-GPU lease owns
    # 16 of its cores (OMP_NUM_THREADS is set to that share), so the share is what "all cores" means there.
    cores = os.cpu_count() or 1
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    if share > 0:
        cores = min(cores, share)
    torch.set_num_threads(cores)
    rng = np.random.default_rng(0)
    offsets = [0]
    for w in lay_widths:
        offsets.append(offsets[-1] + w)
    is_cat = [w > 1 for w in lay_widths]
    C = len(lay_widths)
    Ns = sample_rows
    X = rng.normal(size=(Ns, D)).astype(np.float32)
    data_m = (rng.uniform(size=(Ns, C)) < 0.8).astype(np.int64)
    W = (np.abs(rng.uniform(size=(C, D))) * 1e-3).astype(np.float32)
    k = max(1, min(k_frac_rows, Ns - 1))
    t0 = time.perf_counter()
    orc.attn_fill_dense(torch.tensor(X), torch.tensor(W), data_m, offsets, k)
    t_fill = time.perf_counter() - t0
    # sampler
    from ddgani_amd.synth import random_generator_state
    sd = random_generator_state(D, T)
    Nsamp = 100_000
    x = torch.rand(Nsamp, D)
    M = torch.tensor(orc.expand_cols((rng.uniform(size=(Nsamp, C)) < 0.8).astype(np.float32), offsets))
    S = orc.schedule(T)
    t0 = time.perf_counter()
    with torch.no_grad():
        eps = torch.randn(Nsamp, D)
        zs = [torch.randn(Nsamp, D) for _ in range(T - 2)]
        masks = [[torch.empty(Nsamp, w_).bernoulli_(0.8).div_(0.8) for w_ in (2 * D, D, 2 * D)] for _ in range(T - 1)]
        out = orc.reverse_sample(sd, S, T, x, M, offsets, is_cat, eps, zs, masks)
        orc.mask_merge(x, M, out, offsets, is_cat)
    t_samp = time.perf_counter() - t0
    per_row_fill_full = (t_fill / Ns) * (N_full / Ns)       # seconds per row at N_full keys (extrapolated)
    per_row_samp = t_samp / Nsamp
    value = 1.0 / (per_row_fill_full + per_row_samp)
    return {"value": value, "unit": "rows/s", "cores": cores, "kind": "port",
            "sample": f"attention fill measured on {Ns}x{Ns} rows ({t_fill:.1f} s = {Ns / t_fill:.0f} rows/s there) and "
                      f"extrapolated linearly in N to {N_full} keys; sampler+merge measured on {Nsamp} rows "
                      f"({t_samp:.1f} s = {Nsamp / t_samp:.0f} rows/s); torch CPU fp32, {cores} threads"}


def other_rooflines(ops, P, xd, Md, lay, T, sched_host, reps=5):
    """The two other kernel classes of the hot path, timed live with HIP events on the launch stream AFTER the timed
    region (they are inside `value` already, as part of the sampler phase): the denoiser's dense layers (MFMA-bound)
    and the fused head + reverse step (HBM-bound), each against its roofline."""
    rows, D = xd.shape
    ev = lambda: torch.cuda.Event(enable_timing=True)
    pre = torch.empty_like(xd)
    x_t = xd.clone()
    ops.denoiser_fwd(P, x_t, 3, lay, seed=1, stream_id=1, out=pre, apply_head=False)
    e0, e1, e2 = ev(), ev(), ev()
    e0.record()
    for i in range(reps):
        ops.denoiser_fwd(P, x_t, 3, lay, seed=1, stream_id=2 + i, out=pre, apply_head=False)
    e1.record()
    for i in range(reps):
        ops.head_reverse_step(pre, x_t, xd, Md, 3, T, lay, seed=1, stream_id=50 + i, out=x_t, sched_host=sched_host)
    e2.record()
    torch.cuda.synchronize()
    t_gemm = e0.elapsed_time(e1) / reps * 1e-3            # one denoiser call = 4 dense layers (+ 3 BatchNorm finalizes)
    t_ew = e1.elapsed_time(e2) / reps * 1e-3
    flops = 16.0 * D * D * rows                           # 16 D^2 flop per row and denoiser call
    split = rows >= 8192
    byts = 5.0 * 4.0 * rows * D                           # y, x_t, x, M in; x_{t-1} out (noise on chip)
    return {
        "gemm": {"kernel": "k_linear_wide (4 dense layers of one denoiser call; f16 MFMA on operands split in two halves, "
                           "fp32-class)" if split else "k_linear (fp32 MFMA)",
                 "bound": "mfma", "achieved": flops / t_gemm / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                 "frac": flops / t_gemm / 1e12 / PEAK_F32_MFMA_TFLOPS, "launch_ms": t_gemm * 1e3,
                 "note": "algorithmic fp32 flops against the fp32-matrix peak; the split kernel executes 3 f16 products per "
                         "multiply-add: executed rate %.0f TFLOP/s = %.3f of the 2500 TFLOP/s dense f16 peak"
                         % (3 * flops / t_gemm / 1e12, 3 * flops / t_gemm / 1e12 / PEAK_BF16_MFMA_TFLOPS)},
        "elementwise": {"kernel": "k_head_reverse (per-field head + reverse step, one pass)", "bound": "hbm",
                        "achieved": byts / t_ew / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": byts / t_ew / 1e9 / 8000.0,
                        "launch_ms": t_ew * 1e3, "bytes_per_launch": byts},
    }


def pmc_traffic(workload, rows, kernel_prefix):
    """HBM bytes per step (summed over the kernel's launches of one step, like `achieved`) of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/rNN/pmc_traffic_<workload>_rows<N>.json: FETCH_SIZE and WRITE_SIZE collected in separate
    runs of this same command and corrected as MI355X_MICROARCH.md prescribes).  Counters cannot be read
    from inside a timed run, so the figure is attached only when workload and row count match."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_traffic_{workload}_rows{rows}.json")))[::-1]:
        try:
            d = json.load(open(f))
        except Exception:
            continue
        tot = 0.0
        hit = False
        for k, v in d.get("kernels", {}).items():
            if k.startswith(kernel_prefix) and "fetch_bytes_per_step" in v and "write_bytes_per_step" in v:
                tot += v["fetch_bytes_per_step"] + v["write_bytes_per_step"]
                hit = True
        if hit:
            return tot, os.path.relpath(f, ROOT)
    return None, None


def spawn_ranks(a):
    """`python bench.py --gpus N` from a plain shell: start N ranks (one per GPU) with torch.distributed.run as a CHILD
    process -- before this process has touched the GPU -- relay rank 0's JSON line and exit with the child's code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "16")          # torch.distributed.run would otherwise force 1 (cpu_baseline leg)
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus}", file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    from ddgani_amd import ops, synth
    from ddgani_amd import dist as ddist
    from ddgani_amd.model import Diffusion as D_

    wl = synth.WORKLOADS[a.workload]
    N = a.rows or wl["rows"]
    tab = synth.make_table(a.workload, dev, rows=N)      # every rank builds the same table, keeps its shard
    lay, k, T = tab.layout, tab.k, tab.T
    if a.topk > 0:
        k = int(a.topk) if a.topk >= 1 else max(1, int(N * a.topk))
    comm = ddist.Comm()
    # two row partitions: the attention fill's queries by equal WORK (missing cells x observed keys: Region missingness
    # puts every pair into a block of rows), the sampler's rows by the fixed virtual shards (cost per row)
    abounds = ([(0, N)] if not comm.on else ddist.shard_bounds(N, comm.world) if a.even_shards else
               ddist.work_balanced_bounds(tab.m8, comm.world))
    lo, hi = abounds[comm.rank]
    slo, shi = ddist.shard_bounds(N, comm.world)[comm.rank]
    Xq = tab.X_attn[lo:hi].contiguous()                   # (one GPU: the whole table; else taken from the gathered keys)
    mq = tab.m8[lo:hi].contiguous()
    Xk_local = tab.X_attn[slo:shi].contiguous()           # this rank's contribution to the key all-gather: even shards
    mk_local = tab.m8[slo:shi].contiguous()
    xd = tab.x_diff[slo:shi].contiguous()
    Md = tab.M[slo:shi].contiguous()
    # algorithmic work of the similarity: (query pair, key) units, 2*D flop each
    nobs = tab.m8.sum(0).double()                         # observed keys per column
    nmiss_local = (hi - lo) - mq.sum(0).double()
    pairs_local = float((nmiss_local * nobs).sum().item())
    if world > 1:
        del tab.X_attn, tab.m8, tab.x_diff, tab.M
        torch.cuda.empty_cache()
    P = ops.GenParams(synth.random_generator_state(lay.D, T), dev)
    out_code = torch.empty_like(Xq)
    merged = torch.empty_like(xd)
    sched_host = ops.schedule(T)
    amode = ops.ATTN_BF16_FILTER if a.attn_mode == "bf16filter" else ops.ATTN_F32

    ev = lambda: torch.cuda.Event(enable_timing=True)
    sel_ms, fill_ms, prep_ms, samp_ms = [], [], [], []

    def step(i, record):
        e = [ev() for _ in range(5)]
        e[0].record()
        if comm.on:
            counts = [b - a_ for a_, b in ddist.shard_bounds(N, comm.world)]
            Xk = comm.all_gather_rows(Xk_local, counts)
            mk = comm.all_gather_rows(mk_local, counts)
            plan = ops.AttnPlan(Xk[lo:hi], mk[lo:hi], Xk, mk, tab.W, lay, k, q_row_offset=lo, mode=amode)
        else:
            plan = ops.AttnPlan(Xq, mq, Xq, mq, tab.W, lay, k, mode=amode)
        plan.prepare()
        e[1].record()
        plan.select(out=out_code)
        e[2].record()
        plan.fill()
        e[3].record()
        if comm.on:
            x0_hat = ddist.sharded_reverse_sample(comm, ops, P, T, xd, Md, lay, N, seed=i)
        else:
            x0_hat = D_.reverse_sample(P, T, xd, Md, lay, seed=i, sched_host=sched_host)
        ops.mask_merge(xd, Md, x0_hat, lay, out=merged)
        e[4].record()
        if record:
            torch.cuda.synchronize()
            prep_ms.append(e[0].elapsed_time(e[1]))
            sel_ms.append(e[1].elapsed_time(e[2]))
            fill_ms.append(e[2].elapsed_time(e[3]))
            samp_ms.append(e[3].elapsed_time(e[4]))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i, True)
    barrier()
    dt = time.perf_counter() - t0
    ranks_seen = 1
    if world > 1:
        cdev = dev if a.backend == "nccl" else "cpu"
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        one = torch.ones(1, dtype=torch.int64, device=cdev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)         # ranks that actually ran the timed loop
        ranks_seen = int(one.item())
        wmax = torch.tensor([pairs_local], dtype=torch.float64, device=cdev)
        wsum = wmax.clone()
        dist.all_reduce(wmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(wsum, op=dist.ReduceOp.SUM)
        imbalance = float(wmax.item()) * world / max(float(wsum.item()), 1.0)    # largest share of the similarity work / mean share
        if ranks_seen != a.gpus:
            print(f"bench.py: {ranks_seen} ranks took part, --gpus {a.gpus}", file=sys.stderr)
            sys.exit(3)

    if rank == 0:
        ms = dt / a.steps * 1e3
        sel = sum(sel_ms) / len(sel_ms)
        fil = sum(fill_ms) / len(fill_ms)
        large_k = k > ops.MAX_TOPK_REG                                      # window selection in either mode
        radix = large_k and os.environ.get("DDG_ATTN_RADIX") == "1"
        win_f32 = large_k and os.environ.get("DDG_WIN_F32") == "1"
        n_launch = 5 if radix else 1                                        # radix select = 5 launches of the same kernel
        flops = 2.0 * lay.D * pairs_local
        kernel_ms = sel / n_launch
        if large_k and not radix:
            # two full similarity passes (bracket histogram in the select phase, accumulation in the fill phase); the
            # three sampled passes of the select phase (about 1 % of the keys each) are in the time, not in the flops
            kernel_ms = (sel + fil) / 2
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        filt = amode == ops.ATTN_BF16_FILTER and not large_k
        kname = ("void k_attn_screen<", "void k_attn_rescore<", "void k_attn_warm<") if filt else "void k_attn<"
        # split passes execute three f16 products per algorithmic multiply-add: algorithmic flops against a third of the f16 peak
        peak = PEAK_BF16_MFMA_TFLOPS if filt else (PEAK_BF16_MFMA_TFLOPS / 3 if large_k and not radix and not win_f32 else PEAK_F32_MFMA_TFLOPS)
        traffic, traffic_src = pmc_traffic(a.workload, N, kname) if world == 1 else (None, None)
        res = {
            "metric": "imputed rows/sec (T-step reverse diffusion + top_k attn fill)",
            "value": N / (dt / a.steps), "unit": "rows/s", "n_gpus": world, "n_ranks_seen": ranks_seen, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": ("f32 (f16 MFMA screening with a rigorous margin, exact fp32 re-scoring of every candidate)" if filt else
                      "f32 (logits on the f16 MFMA from operands split in two halves, error 3 * 2^-22 sum|q||k|; fp32 accumulation)"
                      if large_k and not radix and not win_f32 else "f32"),
            "data": "synthetic",
            "config": {"workload": f"{a.workload}: {wl['desc']}" + (f" [rows overridden to {N}]" if a.rows else ""),
                       "rows": N, "cols": lay.C, "encoded_dim": lay.D, "top_k": k, "T": T,
                       "parallelism": (f"rows sharded x{world} (attention queries by " + ("even row counts" if a.even_shards else "equal work")
                                       + f"; largest / mean share of the similarity work {imbalance:.3f})") if world > 1 else "single GPU"},
            "phases_ms": {"prepare(+allgather)": sum(prep_ms) / len(prep_ms), "select": sel,
                          "fill": sum(fill_ms) / len(fill_ms), "sampler+merge": sum(samp_ms) / len(samp_ms)},
            "roofline": {"bound": "mfma",
                         "kernel": ("k_attn_screen (f16 MFMA similarity screening, power-of-two scaled operands; its time here includes the exact fp32 "
                                    "k_attn_rescore and k_attn_warm launches of the select phase, about 12 % of it)" if filt else
                                    "k_attn<MODE_HIST> (fp32 MFMA similarity, 5-level radix select of the k-th logit)" if radix else
                                    "k_attn<MODE_WIN> + k_attn<MODE_WFILL> (window selection: two full similarity passes, "
                                    + ("fp32 MFMA)" if win_f32 else "f16 MFMA on split operands; peak = dense f16 peak / 3 products)") if large_k else
                                    "k_attn<MODE_TOPK> (fp32 MFMA similarity + top-k + fill)"),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "flops_per_launch": flops, "launch_ms": kernel_ms,
                         "launches_note": ("one 'launch' = all key-chunk x workgroup-group launches of the select phase of one "
                                           "step (k_attn_warm once per group, then k_attn_screen + k_attn_rescore per chunk), "
                                           "timed together with HIP events on the launch stream; flops count only the "
                                           "2*D*pairs*keys of the similarity") if filt else
                                          "launch_ms = (select + fill phases) / 2: one full pass each" if large_k and not radix else "one launch per step"},
        }
        if world == 1:
            res["roofline_other_kernels"] = other_rooflines(ops, P, xd, Md, lay, T, sched_host)
            # SURVEY 8(d) "plus (separately) training steps/s": the B = 128 generator + discriminator steps on THIS table at
            # its full row count, after the timed region (not part of `value`)
            import bench_train
            ep = 50
            tdt = bench_train.measure(tab, dev, comm, ep)
            res["training"] = {"metric": "training steps/s (generator + discriminator minibatch steps, B=128, Adam after each)",
                               "value": ep * 2 * T / tdt, "unit": "steps/s", "ms_per_step": tdt / (ep * 2 * T) * 1e3,
                               "rows": N, "steps": ep * 2 * T, "rng": "chip", "graph_replay": os.environ.get("DDG_TRAIN_GRAPH", "1") != "0"}
        if not a.no_cpu and world == 1:
            res["cpu_baseline"] = cpu_baseline(wl, lay.widths, k, T, lay.D, N, a.cpu_rows)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
