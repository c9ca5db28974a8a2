"""BASELINE.md section 3 check: the oracle ("port", what bench.py's cpu_baseline times) against the reference itself at
equal shape on this container's CPU cores.  Run here (the reference tree does not travel to the GPU box):

    python tests/golden/time_port_vs_reference.py            # prints one JSON object; paste into BASELINE.md

Attention fill: the reference's init_attn_2 (utils/util.py:386-501, whole function: encode, N x N similarity per column,
softmax, k-th value, fill, decode) against oracle.attn_fill_dense on the encoded inputs of the same table.
Sampler: the reference's reverse_diffusion_G_D_sample (model/Diffusion.py:426-450) against oracle.reverse_sample +
sampler_disc_tail with the same draws made explicit.
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                      # noqa: E402  (imports the reference, stubs torchmetrics)

orc, ref_util, ref_diff = mg.orc, mg.ref_util, mg.ref_diff


def best(f, n=3):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def attention(N, C, top_k, threads):
    torch.set_num_threads(threads)
    rng = np.random.default_rng(1)
    df, data_m = mg.make_table(rng, N, [0] * C, 0.2)
    corr = mg.corr_like(rng, C)
    t_ref = best(lambda: ref_util.init_attn_2(corr.copy(), df.copy(), data_m.copy(), [], None, [], torch.device("cpu"), top_k))
    code = df.copy()
    code.columns = list(range(C))
    fields, enc_code = ref_util.Data_convert(code, "mean_std", list(range(C)))
    X = torch.tensor(enc_code.values.astype(np.float32))
    offsets = list(range(C + 1))
    W = torch.tensor(orc.expand_cols(corr, offsets).astype(np.float32))
    k = int(N * top_k)
    t_port = best(lambda: orc.attn_fill_dense(X, W, data_m, offsets, k))
    return {"shape": f"{N} rows x {C} numeric columns, top_k {top_k}", "threads": threads,
            "reference_s": round(t_ref, 3), "port_s": round(t_port, 3), "port_over_reference": round(t_port / t_ref, 3)}


def sampler(N, widths, T, threads):
    torch.set_num_threads(threads)
    rng = np.random.default_rng(2)
    G, Dn, fields = mg.build_nets(widths, T)
    G.train(); Dn.train()
    x_np, data_m, M_np = mg.synth_code(rng, N, widths, 0.3)
    x, M = torch.tensor(x_np), torch.tensor(M_np)
    D = x.shape[1]
    offsets, is_cat = mg.layout(widths)
    DS = ref_diff.Diffusion_setting(T, D, torch.device("cpu"))
    PG, PD = mg.sd_t(G), mg.sd_t(Dn)
    S = orc.schedule(T)

    def run_ref():
        with torch.no_grad():
            ref_diff.reverse_diffusion_G_D_sample(G, Dn, DS, T, N, D, torch.device("cpu"), x, M)

    def run_port():
        with torch.no_grad():
            eps_T = torch.randn_like(x)
            zs = [torch.randn_like(x) for _ in range(T - 2)]
            masks = [[torch.empty(N, w).bernoulli_(0.8).div_(0.8) for w in (2 * D, D, 2 * D)] for _ in range(T - 1)]
            eps2 = torch.randn_like(x)
            o = orc.reverse_sample(PG, S, T, x, M, offsets, is_cat, eps_T, zs, masks)
            orc.sampler_disc_tail(PD, S, x, o, eps2)

    t_ref, t_port = best(run_ref), best(run_port)
    return {"shape": f"{N} rows x D {D}, T {T}", "threads": threads, "reference_s": round(t_ref, 3),
            "port_s": round(t_port, 3), "port_over_reference": round(t_port / t_ref, 3)}


if __name__ == "__main__":
    thr = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1)
    res = {"host": f"{os.cpu_count()} CPUs", "attention": [attention(2048, 16, 0.05, thr), attention(4096, 16, 0.05, thr)],
           "sampler": [sampler(100_000, [1, 1, 1, 8, 1, 1, 1, 8] * 4, 6, thr)]}
    print(json.dumps(res, indent=1))
