"""GPU: every HIP entry point, called through the C-ABI, against the oracle / golden vectors.

Tolerances: elementwise passes are bit-exact or within 2 fp32 ulps of the oracle; GEMM-based
results within 1e-5 abs (north star: 1e-4 relative); attention fill: tau / kept set exact modulo
near-ties within 1e-6 of the k-th logit (north star), values within 1e-5 where no near-tie exists.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ddgani_oracle as orc
from util_golden import layout_of, load, sampler_randomness, state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _ops():
    from ddgani_amd import ops
    return ops


# ----------------------------------------------------------------------------- elementwise
@pytest.mark.parametrize("name,T", [("sampler_num", 6), ("sampler_mixed", 6), ("sampler_T10", 10)])
def test_forward_noise_and_reverse_step(dev, name, T):
    ops = _ops()
    z = load(name)
    x, M = torch.tensor(z["x"]).to(dev), torch.tensor(z["M"]).to(dev)
    x_T = ops.forward_noise(x, T, T, eps=torch.tensor(z["eps_T"]).to(dev))
    assert np.array_equal(x_T.cpu().numpy(), z["x_T"])
    out = ops.reverse_step(x_T, x, M, torch.tensor(z["pred_T"]).to(dev), T, T, z=torch.tensor(z["z0"]).to(dev))
    assert np.array_equal(out.cpu().numpy(), z["x_Tm1"])


def test_forward_noise_per_row_t(dev):
    ops = _ops()
    z = load("train_mixed")
    T = int(z["T"])
    x0 = torch.tensor(z["x0"]).to(dev)
    t = torch.tensor(z["t"]).to(dev)
    xt = ops.forward_noise(x0, t, T, eps=torch.tensor(z["eps"]).to(dev))
    # schedule from the C entry point (pinned to the golden schedule in test_abi): torch's vectorised sqrt is
    # not correctly rounded on every host CPU, so orc.schedule() may differ by 1 ulp between machines
    S = {k: torch.tensor(v) for k, v in ops.schedule_dict(T).items()}
    ref = orc.forward_noise(S, torch.tensor(z["x0"]), torch.tensor(z["t"]), torch.tensor(z["eps"]))
    assert np.array_equal(xt.cpu().numpy(), ref.numpy())


def test_onchip_noise_statistics(dev):
    """Philox/Box-Muller noise: mean 0, var 1, reproducible per (seed, stream), distinct across streams."""
    ops = _ops()
    x0 = torch.zeros(4096, 64, device=dev)
    T = 6
    a, ea = ops.forward_noise(x0, T, T, seed=7, stream_id=1, return_eps=True)
    b, eb = ops.forward_noise(x0, T, T, seed=7, stream_id=1, return_eps=True)
    c, ec = ops.forward_noise(x0, T, T, seed=7, stream_id=2, return_eps=True)
    assert torch.equal(ea, eb) and not torch.equal(ea, ec)
    e = ea.double()
    assert abs(e.mean().item()) < 0.01 and abs(e.var().item() - 1) < 0.02
    assert abs((e ** 4).mean().item() - 3) < 0.15
    s = orc.schedule(T)
    assert torch.allclose(a, s["sqrt_one_minus_alpha_cumulative"][T - 1].item() * ea, atol=1e-7)


@pytest.mark.parametrize("name", ["sampler_mixed", "sampler_T10", "sampler_num"])
def test_head_and_mask_merge(dev, name):
    ops = _ops()
    z = load(name)
    lay = ops.FieldLayout(z["widths"].tolist(), dev)
    out = ops.head(torch.tensor(z["pre_T"]).to(dev), lay)
    np.testing.assert_allclose(out.cpu().numpy(), z["pred_T"], rtol=0, atol=3e-7)
    code = ops.mask_merge(torch.tensor(z["x"]).to(dev), torch.tensor(z["M"]).to(dev),
                          torch.tensor(z["x0_hat"]).to(dev), lay)
    np.testing.assert_allclose(code.cpu().numpy(), z["code"], rtol=0, atol=3e-7)


# ----------------------------------------------------------------------------- denoiser / discriminator
@pytest.mark.parametrize("name,T", [("sampler_num", 6), ("sampler_mixed", 6), ("sampler_T10", 10)])
def test_denoiser_forward(dev, name, T):
    ops = _ops()
    z = load(name)
    lay = ops.FieldLayout(z["widths"].tolist(), dev)
    P = ops.GenParams(state_dict(z, "G__"), dev)
    _, masks_b, _ = sampler_randomness(z, T)
    drops = [torch.tensor(m.astype(np.uint8)).to(dev) for m in masks_b[0]]
    x_T = torch.tensor(z["x_T"]).to(dev)
    pre = ops.denoiser_fwd(P, x_T, T, lay, drops=drops, apply_head=False)
    np.testing.assert_allclose(pre.cpu().numpy(), z["pre_T"], rtol=1e-4, atol=2e-5)
    pred = ops.denoiser_fwd(P, x_T, T, lay, drops=drops)
    np.testing.assert_allclose(pred.cpu().numpy(), z["pred_T"], rtol=0, atol=1e-5)


def test_denoiser_per_row_t_and_big_batch(dev):
    """per-row timesteps (training batches) + a row count that is not a multiple of the tile."""
    ops = _ops()
    z = load("train_mixed")
    T = int(z["T"])
    lay = ops.FieldLayout(z["widths"].tolist(), dev)
    sd = state_dict(z, "G0__")
    P = ops.GenParams(sd, dev)
    off, cat = layout_of(z["widths"])
    g = torch.Generator().manual_seed(5)
    N = 1000
    x = torch.rand(N, lay.D, generator=g)
    t = torch.randint(2, T + 1, (N,), generator=g)
    masks = [torch.rand(N, w, generator=g) < 0.8 for w in (2 * lay.D, lay.D, 2 * lay.D)]
    with torch.no_grad():
        ref = orc.generator_forward(sd, x, t, off, cat, [m.float() / 0.8 for m in masks])
    out = ops.denoiser_fwd(P, x.to(dev), t.to(dev), lay, drops=[m.to(torch.uint8).to(dev) for m in masks])
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-5)
    # golden training batch
    masks = [torch.tensor(z[f"mask_{l}"].astype(np.uint8)).to(dev) for l in range(3)]
    xt = ops.forward_noise(torch.tensor(z["x0"]).to(dev), torch.tensor(z["t"]).to(dev), T,
                           eps=torch.tensor(z["eps"]).to(dev))
    pred = ops.denoiser_fwd(P, xt, torch.tensor(z["t"]).to(dev), lay, drops=masks)
    np.testing.assert_allclose(pred.cpu().numpy(), z["pred"], rtol=0, atol=1e-5)


def test_denoiser_onchip_dropout_rate(dev):
    ops = _ops()
    z = load("sampler_num")
    lay = ops.FieldLayout(z["widths"].tolist(), dev)
    P = ops.GenParams(state_dict(z, "G__"), dev)
    x = torch.rand(5000, lay.D, device=dev)
    a = ops.denoiser_fwd(P, x, 3, lay, drop_p=0.2, seed=1, stream_id=3)
    b = ops.denoiser_fwd(P, x, 3, lay, drop_p=0.2, seed=1, stream_id=3)
    c = ops.denoiser_fwd(P, x, 3, lay, drop_p=0.2, seed=2, stream_id=3)
    d = ops.denoiser_fwd(P, x, 3, lay, drop_p=0.0)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.isfinite(a).all() and (a - d).abs().mean() > 1e-4


@pytest.mark.parametrize("name", ["sampler_mixed", "sampler_num"])
def test_discriminator_forward(dev, name):
    ops = _ops()
    z = load(name)
    T = int(z["T"])
    sd = state_dict(z, "D__")
    P = ops.DiscParams(sd, dev)
    g = torch.Generator().manual_seed(3)
    N = 333
    D = P.in_dim // 2
    a, b = torch.randn(N, D, generator=g), torch.randn(N, D, generator=g)
    t = torch.randint(2, T + 1, (N,), generator=g)
    with torch.no_grad():
        ref = orc.disc_forward(sd, a, b, t)
    out = ops.disc_fwd(P, a.to(dev), b.to(dev), t.to(dev))
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-6)


@pytest.mark.parametrize("name,T", [("sampler_num", 6), ("sampler_mixed", 6), ("sampler_T10", 10)])
def test_full_sampler_against_reference(dev, name, T):
    """whole T-step reverse diffusion with the reference's explicit random draws (golden x0_hat)."""
    from ddgani_amd.model import Diffusion as D_
    ops = _ops()
    z = load(name)
    lay = ops.FieldLayout(z["widths"].tolist(), dev)
    P = ops.GenParams(state_dict(z, "G__"), dev)
    zs, masks_b, _ = sampler_randomness(z, T)
    x, M = torch.tensor(z["x"]).to(dev), torch.tensor(z["M"]).to(dev)
    rnd = D_.ExplicitRandomness(eps_T=torch.tensor(z["eps_T"]).to(dev), zs=[t.to(dev) for t in zs],
                                drops=[[torch.tensor(m.astype(np.uint8)).to(dev) for m in ms] for ms in masks_b])
    x0_hat = D_.reverse_sample(P, T, x, M, lay, randomness=rnd)
    np.testing.assert_allclose(x0_hat.cpu().numpy(), z["x0_hat"], rtol=0, atol=2e-5)
    code = ops.mask_merge(x, M, x0_hat, lay)
    np.testing.assert_allclose(code.cpu().numpy(), z["code"], rtol=0, atol=2e-5)


# ----------------------------------------------------------------------------- attention fill
def _attn_case(dev, X, W, data_m, widths, k, mode=1):
    ops = _ops()
    lay = ops.FieldLayout(widths, dev)
    Mx = orc.expand_cols(data_m.astype(np.float32), lay.offsets)
    X0 = torch.tensor(X * Mx).to(dev)
    m8 = torch.tensor(data_m.astype(np.uint8)).to(dev)
    out, tau, nval = ops.attn_topk_fill(X0, m8, X0, m8, torch.tensor(W).to(dev), lay, k, debug=True, mode=mode)
    torch.cuda.synchronize()
    return out.cpu().numpy(), tau.cpu().numpy(), nval.cpu().numpy(), lay


def _check_attn(out, tau, nval, lay, X, W, data_m, k, ref_code=None):
    off = lay.offsets
    cf, info = orc.attn_fill_closed_form(torch.tensor(X), torch.tensor(W), data_m, off, k, tie_tol=1e-6)
    obs = np.repeat(data_m, np.diff(off), axis=1) == 1
    Mx = obs.astype(np.float32)
    assert np.array_equal(out[obs], (X * Mx)[obs])           # observed cells untouched
    nflag = 0
    for (j, c), d in info.items():
        a, b = off[c], off[c + 1]
        assert nval[j, c] == d["n_valid"], (j, c)
        if np.isfinite(d["tau"]):
            assert abs(tau[j, c] - d["tau"]) <= 1e-6, (j, c, tau[j, c], d["tau"])   # k-th logit: set exact mod near-ties
        else:
            assert tau[j, c] == -np.inf
        err = np.abs(out[j, a:b] - cf[j, a:b]).max()
        if d["near_tie"]:
            nflag += 1
            assert err < 1.0
        else:
            assert err < 1e-5, (j, c, err)
            if ref_code is not None:
                assert np.abs(out[j, a:b] - ref_code[j, a:b]).max() < 1e-5, (j, c)
    assert nflag <= 0.1 * max(len(info), 1)
    # cells that are observed report NaN tau
    assert np.isnan(tau[data_m == 1]).all()


MODES = [pytest.param(0, id="f32"), pytest.param(1, id="bf16filter")]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["attn_num64", "attn_mixed257", "attn_onehot200", "attn_num600", "attn_adult2000"])
def test_attn_fill_golden(dev, name, mode):
    """fractional-k fixtures from the reference's init_attn_2 (k = int(N*top_k))."""
    z = load(name)
    widths = np.diff(z["offsets"]).tolist()
    k = int(z["k"])
    out, tau, nval, lay = _attn_case(dev, z["X"], z["W"], z["data_m"], widths, k, mode)
    _check_attn(out, tau, nval, lay, z["X"], z["W"], z["data_m"], k, ref_code=z["new_code"])


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("N,widths,k,miss", [
    (300, [1] * 20, 5, 0.3),          # register top-k (KCAP 16)
    (500, [1, 4, 1, 6, 1, 1, 3], 20, 0.4),   # register top-k (KCAP 32), categorical blocks
    (700, [1] * 70, 50, 0.5),         # radix select, D in (64,128]
    (400, [1] * 150, 10, 0.5),        # D in (128,192]
    (300, [8] * 30, 7, 0.2),          # D = 240 one-hot heavy, massive ties
    (260, [1] * 9, 400, 0.3),         # k larger than the number of valid keys: keep all
    (130, [1, 1, 1], 3, 0.9),         # very sparse: some columns have < k or 0 valid keys
    (350, [1] * 12, 14, 0.3),         # largest k whose k+2 list fits 16 registers
    (350, [1] * 12, 15, 0.3),         # smallest k on the 32-register lists
    (350, [1] * 12, 30, 0.3),         # DDG_MAX_TOPK_REG
    (350, [1] * 12, 31, 0.3),         # first k on the radix path
])
def test_attn_fill_synthetic(dev, N, widths, k, miss, mode):
    rng = np.random.default_rng(N + k)
    cols = []
    for w in widths:
        if w == 1:
            cols.append(rng.normal(size=(N, 1)))
        else:
            oh = np.zeros((N, w))
            oh[np.arange(N), rng.integers(0, w, N)] = 1
            cols.append(oh)
    X = np.concatenate(cols, 1).astype(np.float32)
    C = len(widths)
    data_m = (rng.uniform(size=(N, C)) < 1 - miss).astype(np.int64)
    if miss >= 0.9:
        data_m[:, 0] = 0
        data_m[0, 0] = 1                      # column 0: a single valid key
        data_m[:, 1] = 0                      # column 1: no valid key at all -> fill 0
    off, _ = layout_of(widths)
    W = orc.expand_cols(np.abs(rng.uniform(size=(C, C))) * (1e-3 if N % 2 == 0 else 0.05), off).astype(np.float32)
    out, tau, nval, lay = _attn_case(dev, X, W, data_m, widths, k, mode)
    _check_attn(out, tau, nval, lay, X, W, data_m, k)


@pytest.mark.parametrize("arith", ["split", "f32"])
@pytest.mark.parametrize("N,widths,k,miss", [
    (700, [1] * 70, 50, 0.5),         # D in (64,128]
    (600, [8] * 20, 40, 0.3),         # one-hot heavy: hundreds of exactly tied logits around the k-th
    (501, [1, 12, 1, 6, 1, 1, 3], 45, 0.4),   # a block wider than the side tile (d_c > 8)
    (260, [1] * 9, 400, 0.3),         # k larger than the number of valid keys: keep all
    (400, [1] * 10, 300, 0.2),        # k close to the number of valid keys: the bracket has no lower edge
    (350, [1] * 12, 31, 0.3),         # first k of the path: no upper edge
    (130, [1, 1, 1], 40, 0.9),        # very sparse: columns with 0 or 1 valid keys
])
def test_attn_fill_synthetic_window_path(dev, N, widths, k, miss, arith):
    """the k > DDG_MAX_TOPK_REG cases of test_attn_fill_synthetic through the window selection (forced: at these sizes
    the library would take the radix path), against the fp64 closed form."""
    os.environ["DDG_ATTN_WINDOW"] = "1"
    if arith == "f32":
        os.environ["DDG_WIN_F32"] = "1"
    try:
        test_attn_fill_synthetic(dev, N, widths, k, miss, 0)
    finally:
        os.environ.pop("DDG_ATTN_WINDOW", None)
        os.environ.pop("DDG_WIN_F32", None)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k", [1, 4, 9])
def test_attn_fill_duplicate_keys(dev, mode, k):
    """exact ties at the k-th logit from duplicated rows: two-way ties are resolved inside the register
    path (both keys kept, as utils/util.py:470-476 keeps every probability >= the k-th), three-way ties
    go through the threshold pass; all must agree with the closed form bit-for-set."""
    rng = np.random.default_rng(77 + k)
    base = rng.normal(size=(120, 10)).astype(np.float32)
    mb = (rng.uniform(size=(120, 10)) < 0.7).astype(np.int64)
    uniq = rng.normal(size=(60, 10)).astype(np.float32)
    mu = (rng.uniform(size=(60, 10)) < 0.7).astype(np.int64)
    X = np.concatenate([base, base, base[:60], uniq], 0)       # rows 0-59 three times, 60-119 twice
    data_m = np.concatenate([mb, mb, mb[:60], mu], 0)
    widths = [1] * 10
    off, _ = layout_of(widths)
    W = orc.expand_cols(np.abs(rng.uniform(size=(10, 10))) * 0.05, off).astype(np.float32)
    out, tau, nval, lay = _attn_case(dev, X, W, data_m, widths, k, mode)
    _check_attn(out, tau, nval, lay, X, W, data_m, k)


@pytest.mark.parametrize("mode", MODES)
def test_attn_fill_sharded_queries(dev, mode):
    """query rows = a shard of the key rows (multi-GPU layout): same result as the full call."""
    ops = _ops()
    z = load("attn_mixed257")
    widths = np.diff(z["offsets"]).tolist()
    lay = ops.FieldLayout(widths, dev)
    Mx = orc.expand_cols(z["data_m"].astype(np.float32), lay.offsets)
    X0 = torch.tensor(z["X"] * Mx).to(dev)
    m8 = torch.tensor(z["data_m"].astype(np.uint8)).to(dev)
    W = torch.tensor(z["W"]).to(dev)
    k = int(z["k"]) if mode == 0 else 7      # 7: register path, so that mode 1 runs the bf16 screening kernel
    full = ops.attn_topk_fill(X0, m8, X0, m8, W, lay, k, mode=mode)
    lo, hi = 100, 257
    part = ops.attn_topk_fill(X0[lo:hi].contiguous(), m8[lo:hi].contiguous(), X0, m8, W, lay, k, q_row_offset=lo,
                              mode=mode)
    assert torch.equal(part, full[lo:hi])


def _assert_modes_agree(tab, lo, res_a, res_b, dev, max_checked=12):
    """two results (out, tau) for the query shard starting at row lo: tau within 1e-6 everywhere; fills equal to fp32
    summation order except on pairs that an fp64 recomputation on the host shows to be near-ties (the k-th and
    (k+1)-th logit closer than 1e-6: membership legitimately depends on the fp32 summation order there)."""
    (oa, ta), (ob, tb) = res_a, res_b
    assert torch.equal(torch.isnan(ta), torch.isnan(tb))
    dt = (ta - tb).abs()
    fin = ~torch.isnan(dt)
    assert float(dt[fin].max()) <= 1e-6
    widths = torch.tensor(tab.layout.widths, device=dev)
    diff = (oa - ob).abs()
    col_of = torch.repeat_interleave(torch.arange(len(tab.layout.widths), device=dev), widths)
    bad = torch.nonzero(diff >= 2e-6)
    pairs = sorted({(int(j), int(col_of[cc])) for j, cc in bad.tolist()})
    assert len(pairs) <= 1e-3 * max(int(fin.sum()), 1000), len(pairs)
    if pairs:
        X64 = tab.X_attn.double()
        for j, c in pairs[:max_checked]:
            keys = torch.nonzero(tab.m8[:, c] == 1)[:, 0]
            keys = keys[keys != lo + j]
            s = X64[keys] @ (tab.W[c].double() * X64[lo + j])
            top = torch.topk(s, tab.k + 1).values
            assert float(top[tab.k - 1] - top[tab.k]) <= 1e-6, (j, c, top[-3:].tolist())


def test_attn_full_size_modes_agree_and_match_fp64(dev):
    """BASELINE configuration c3 at its full size (1M key rows x 64 columns, D = 176, k = 10): a shard of 1 536
    query rows is filled against ALL keys by the bf16-screened path (two key-chunk schedules) and by the fp32
    path; the three results must agree to fp32 summation order, and spot pairs must match an fp64 closed form
    computed here on the host (tau within 1e-6, values within 1e-5, near-ties within 1e-6 of tau exempt)."""
    ops = _ops()
    from ddgani_amd import synth
    tab = synth.make_table("c3", dev)
    N = tab.X_attn.shape[0]
    lo, hi = 300_000, 301_536
    Xq, mq = tab.X_attn[lo:hi].contiguous(), tab.m8[lo:hi].contiguous()
    outs, taus = [], []
    for mode, chunk in ((1, None), (1, 40000), (0, None)):
        if chunk:
            ops.attn_set_key_chunk(chunk)
        try:
            o, t, nv = ops.attn_topk_fill(Xq, mq, tab.X_attn, tab.m8, tab.W, tab.layout, tab.k, q_row_offset=lo,
                                          debug=True, mode=mode)
            torch.cuda.synchronize()
        finally:
            ops.attn_set_key_chunk(-1)
        outs.append(o)
        taus.append(t)
    for o, t in zip(outs[:2], taus[:2]):
        _assert_modes_agree(tab, lo, (o, t), (outs[2], taus[2]), dev)
    # fp64 closed form for a handful of (query, column) pairs
    X64 = tab.X_attn.double().cpu().numpy()
    m = tab.m8.cpu().numpy()
    W64 = tab.W.double().cpu().numpy()
    off = tab.layout.offsets
    out = outs[0].cpu().numpy()
    tau = taus[0].cpu().numpy()
    rng = np.random.default_rng(5)
    checked = 0
    for j in rng.choice(hi - lo, 6, replace=False):
        for c in np.where(m[lo + j] == 0)[0][:6]:
            keys = np.where(m[:, c] == 1)[0]
            keys = keys[keys != lo + j]
            s = X64[keys] @ (W64[c] * X64[lo + j])
            kth = np.partition(s, len(s) - tab.k)[len(s) - tab.k]
            assert abs(tau[j, c] - kth) <= 1e-6, (j, c)
            if np.any((s < kth) & (s >= kth - 1e-6)):
                continue                                   # near-tie: membership may differ, exempt
            keep = s >= kth
            w = np.exp(s[keep] - s.max())
            ref = (w / w.sum()) @ X64[keys[keep], off[c]:off[c + 1]]
            assert np.abs(out[j, off[c]:off[c + 1]] - ref).max() < 1e-5, (j, c)
            checked += 1
    assert checked >= 20


@pytest.mark.parametrize("wl,rows", [("c4", 400_000), ("c5", 300_000)])
def test_attn_other_configs_modes_agree(dev, wl, rows):
    """BASELINE configurations c4 (128 numeric columns, MCAR 0.5: 8 MFMA k-steps) and c5 (256 numeric columns, Region
    missingness: 16 k-steps, 64-key stages) at a row count one GPU test can afford: a shard of 1 024 query rows
    against all keys, bf16-screened path vs fp32 path (same property as the c3 full-size test)."""
    ops = _ops()
    from ddgani_amd import synth
    tab = synth.make_table(wl, dev, rows=rows)
    miss_rows = torch.nonzero((tab.m8 == 0).any(1))[:, 0]
    assert miss_rows.numel() > 0
    lo = int(miss_rows[miss_rows.numel() // 2])            # a region that has missing cells (Region masks are local)
    lo = max(0, min(lo - 512, rows - 1024))
    hi = lo + 1024
    Xq, mq = tab.X_attn[lo:hi].contiguous(), tab.m8[lo:hi].contiguous()
    assert int((mq == 0).sum()) > 0
    res = []
    for mode in (1, 0):
        o, t, nv = ops.attn_topk_fill(Xq, mq, tab.X_attn, tab.m8, tab.W, tab.layout, tab.k, q_row_offset=lo,
                                      debug=True, mode=mode)
        torch.cuda.synchronize()
        res.append((o, t, nv))
    (o1, t1, n1), (o0, t0, n0) = res
    assert torch.equal(n1, n0)
    _assert_modes_agree(tab, lo, (o1, t1), (o0, t0), dev)


# ----------------------------------------------------------------------------- split-precision layers (large row counts)
def test_linear_split_kernel_is_fp32_class(dev):
    """From 8 192 rows on, ddg_linear runs on the f16 matrix pipe with both operands split in two halves
    (csrc/gemm.hip: k_linear_split; error <= ~3 * 2^-22 * sum|a||w|).  Against an fp64 product: every prologue, the
    concatenated operand of the discriminator, widths that are not multiples of the tile, tiny and large magnitudes;
    and the fp32 kernel (below the row threshold) on the same rows for comparison."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    N = 20000
    for (din, dout) in ((176, 352), (352, 176), (57, 114), (290, 72)):
        A = torch.randn(N, din, generator=g) * torch.logspace(-4, 1, din)          # columns from 1e-4 to 10
        W = (torch.rand(dout, din, generator=g) - 0.5) * 0.15
        b = torch.randn(dout, generator=g) * 0.1
        ref = A.double() @ W.double().T + b.double()
        bound = (A.double().abs() @ W.double().abs().T)                             # sum |a||w| per output
        got = ops.linear(A.to(dev), W.to(dev), b.to(dev)).cpu().double()
        err = ((got - ref).abs() / bound.clamp_min(1e-30)).max().item()
        assert err < 1.5e-6, (din, dout, err)                                       # 3 * 2^-22 = 7e-7 plus fp32 accumulation
        small = ops.linear(A[:4096].to(dev), W.to(dev), b.to(dev)).cpu().double()   # fp32 kernel
        assert (small - got[:4096]).abs().max().item() < 2e-6 * bound.max().item()
        # column statistics of the split kernel's output
        stats = torch.zeros(2 * dout, dtype=torch.float64, device=dev)
        y = ops.linear(A.to(dev), W.to(dev), b.to(dev), stats=stats)
        assert torch.allclose(stats[:dout].cpu(), y.double().sum(0).cpu(), rtol=1e-12, atol=1e-9)
        assert torch.allclose(stats[dout:].cpu(), (y.double() ** 2).sum(0).cpu(), rtol=1e-12, atol=1e-9)
    # concatenated operand + per-row embedding prologue (discriminator layers) at large N
    z = load("sampler_mixed")
    sd = state_dict(z, "D__")
    P = ops.DiscParams(sd, dev)
    D = P.in_dim // 2
    a, b2 = torch.randn(N, D, generator=g), torch.randn(N, D, generator=g)
    t = torch.randint(2, int(z["T"]) + 1, (N,), generator=g)
    with torch.no_grad():
        ref = orc.disc_forward(sd, a, b2, t)
    out = ops.disc_fwd(P, a.to(dev), b2.to(dev), t.to(dev))
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-6)


def test_denoiser_large_rows_split_path(dev):
    """the whole denoiser (BatchNorm / ReLU / dropout prologues) at 20 000 rows -- the split kernels -- against the
    oracle, same bar as the small fixtures (1e-5)"""
    ops = _ops()
    z = load("train_mixed")
    T = int(z["T"])
    lay = ops.FieldLayout(z["widths"].tolist(), dev)
    sd = state_dict(z, "G0__")
    P = ops.GenParams(sd, dev)
    off, cat = layout_of(z["widths"])
    g = torch.Generator().manual_seed(6)
    N = 20000
    x = torch.rand(N, lay.D, generator=g)
    t = torch.randint(2, T + 1, (N,), generator=g)
    masks = [torch.rand(N, w, generator=g) < 0.8 for w in (2 * lay.D, lay.D, 2 * lay.D)]
    with torch.no_grad():
        ref = orc.generator_forward(sd, x, t, off, cat, [m.float() / 0.8 for m in masks])
    out = ops.denoiser_fwd(P, x.to(dev), t.to(dev), lay, drops=[m.to(torch.uint8).to(dev) for m in masks])
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-5)


def test_attn_large_k_at_scale(dev):
    """The reference's own top_k semantics (param.json: top_k = 0.05, k = int(0.05 N)) at a size where k is in the
    thousands: c3-shaped table of 200 000 rows, k = 10 000, a shard of 512 query rows against all keys (radix select
    of the k-th logit + threshold fill; both modes take this path for k > DDG_MAX_TOPK_REG).  Spot pairs against an
    fp64 closed form: tau within 1e-6; the fill within 1e-5 plus the weight of the keys whose logit lies within 1e-6
    of tau (with k in the thousands such near-ties exist for every pair; the north star exempts their membership)."""
    ops = _ops()
    from ddgani_amd import synth
    rows, k = 200_000, 10_000
    tab = synth.make_table("c3", dev, rows=rows)
    lo, hi = 77_000, 77_512
    Xq, mq = tab.X_attn[lo:hi].contiguous(), tab.m8[lo:hi].contiguous()
    res = []
    for mode in (1, 0):
        o, t, nv = ops.attn_topk_fill(Xq, mq, tab.X_attn, tab.m8, tab.W, tab.layout, k, q_row_offset=lo, debug=True,
                                      mode=mode)
        torch.cuda.synchronize()
        res.append((o, t, nv))
    (o1, t1, n1), (o0, t0, n0) = res
    assert torch.equal(n1, n0) and torch.equal(torch.isnan(t1), torch.isnan(t0))
    fin = ~torch.isnan(t1)
    assert int(fin.sum()) > 1000
    assert float((t1 - t0).abs()[fin].max()) <= 1e-6
    X64 = tab.X_attn.double().cpu().numpy()
    m = tab.m8.cpu().numpy()
    W64 = tab.W.double().cpu().numpy()
    off = tab.layout.offsets
    rng = np.random.default_rng(9)
    checked = 0
    with_missing = np.where((m[lo:hi] == 0).any(1))[0]
    for j in rng.choice(with_missing, 8, replace=False):
        for c in np.where(m[lo + j] == 0)[0][:3]:
            keys = np.where(m[:, c] == 1)[0]
            keys = keys[keys != lo + j]
            s = X64[keys] @ (W64[c] * X64[lo + j])
            kth = np.partition(s, len(s) - k)[len(s) - k]
            xv = X64[keys, off[c]:off[c + 1]]
            keep = s >= kth
            w = np.exp(s - s.max())
            Z = w[keep].sum()
            ref = (w[keep] @ xv[keep]) / Z
            band = np.abs(s - kth) <= 1e-6
            slack = (w[band, None] * np.abs(xv[band] - ref)).sum(0) / Z + 1e-5
            for o, t in ((o1, t1), (o0, t0)):
                assert abs(float(t[j, c]) - kth) <= 1e-6, (j, c, float(t[j, c]), kth)
                got = o[j, off[c]:off[c + 1]].double().cpu().numpy()
                assert np.all(np.abs(got - ref) <= slack), (j, c, np.abs(got - ref).max(), slack.max())
            checked += 1
    assert checked >= 12


@pytest.mark.parametrize("arith", ["split", "f32"])
@pytest.mark.parametrize("rows,k,cap", [
    (40_000, 2_000, None),        # top_k = 0.05: sampled bracket (stride 2), two full passes
    (40_000, 31, None),           # smallest k of the path: the bracket has no upper edge
    (40_000, 28_000, None),       # most keys kept
    (40_000, 2_000, "3"),         # lists of 3 entries: most pairs overflow into the radix fallback
    (3_000, 150, None),           # fewer keys than the sample size: the "sample" is every key
])
def test_attn_window_path_equals_radix(dev, rows, k, cap, arith):
    """k > DDG_MAX_TOPK_REG: the window selection (sample bracket -> one histogram pass -> accumulate + exact in-list
    selection) against the 5-level radix select of round 1.  With DDG_WIN_F32=1 both run the same fp32 MFMA chain: the
    k-th logit must be the SAME float, the fills equal up to summation order.  By default the window passes compute
    the logits on the f16 MFMA from operands split in two halves (error <= 3 * 2^-22 sum|q||k|, here < 2e-7): the k-th
    logit within that, the fills within 1e-5 apart from pairs whose k-th and (k+1)-th logits are closer than the
    arithmetic's error (north star: membership of keys within 1e-6 of the k-th logit is exempt).  The radix path itself
    is held to the fp64 closed form in test_attn_large_k_at_scale / test_attn_fill_synthetic."""
    ops = _ops()
    from ddgani_amd import synth
    tab = synth.make_table("c3", dev, rows=rows)
    lo, hi = rows // 3, rows // 3 + 384
    Xq, mq = tab.X_attn[lo:hi].contiguous(), tab.m8[lo:hi].contiguous()
    res = {}
    for radix in ("1", "0"):
        os.environ["DDG_ATTN_RADIX"] = radix
        if radix == "0":
            os.environ["DDG_ATTN_WINDOW"] = "1"       # also below the size at which the library switches to it
            if cap:
                os.environ["DDG_WIN_CAP"] = cap
            if arith == "f32":
                os.environ["DDG_WIN_F32"] = "1"
        try:
            o, t, nv = ops.attn_topk_fill(Xq, mq, tab.X_attn, tab.m8, tab.W, tab.layout, k, q_row_offset=lo, debug=True,
                                          mode=0)
            torch.cuda.synchronize()
        finally:
            for v in ("DDG_ATTN_RADIX", "DDG_WIN_CAP", "DDG_WIN_F32", "DDG_ATTN_WINDOW"):
                os.environ.pop(v, None)
        res[radix] = (o, t, nv)
    (o1, t1, n1), (o0, t0, n0) = res["1"], res["0"]
    assert torch.equal(n1, n0)
    fin = ~torch.isnan(t1)
    assert torch.equal(torch.isnan(t0), torch.isnan(t1)) and int(fin.sum()) > 1000
    if arith == "f32":
        assert torch.equal(t0[fin], t1[fin])                              # the same k-th logit, bit for bit
        assert float((o0 - o1).abs().max()) <= 2e-6
    else:
        assert float((t0[fin] - t1[fin]).abs().max()) <= 2e-7
        d = (o0 - o1).abs()
        lay_off = tab.layout.offsets
        cell = torch.stack([d[:, lay_off[c]:lay_off[c + 1]].max(1).values for c in range(len(lay_off) - 1)], 1)
        bad = (cell > 1e-5) & fin
        assert int(bad.sum()) <= 0.02 * int(fin.sum()), (int(bad.sum()), int(fin.sum()))
        assert float(cell[fin].max()) < 2.0 / k + 1e-3      # a flipped near-tie moves one key of k in or out


def _fp64_spot_pairs(tab, lo, mq_h, o, t, k, n_rows, per_row, seed):
    """(row, column) pairs of a query shard against the fp64 closed form: the k-th logit within 1e-6, the fill within
    1e-5 plus the weight of the keys whose logit lies within 1e-6 of it (near-ties: exempt by the north star)"""
    X64, W64, m, off = tab.X_attn.double(), tab.W.double(), tab.m8, tab.layout.offsets
    rng = np.random.default_rng(seed)
    with_missing = np.where((mq_h == 0).any(1))[0]
    checked = 0
    for j in rng.choice(with_missing, n_rows, replace=False):
        for c in np.where(mq_h[j] == 0)[0][:per_row]:
            keys = torch.nonzero(m[:, c] == 1).flatten()
            keys = keys[keys != lo + j]
            s = (X64[keys] @ (W64[c] * X64[lo + j])).cpu().numpy()
            kth = np.partition(s, len(s) - k)[len(s) - k]
            xv = X64[keys][:, off[c]:off[c + 1]].cpu().numpy()
            keep = s >= kth
            w = np.exp(s - s.max())
            Z = w[keep].sum()
            ref = (w[keep] @ xv[keep]) / Z
            band = np.abs(s - kth) <= 1e-6
            slack = (w[band, None] * np.abs(xv[band] - ref)).sum(0) / Z + 1e-5
            assert abs(float(t[j, c]) - kth) <= 1e-6, (j, c, float(t[j, c]), kth)
            got = o[j, off[c]:off[c + 1]].double().cpu().numpy()
            assert np.all(np.abs(got - ref) <= slack), (j, c, np.abs(got - ref).max(), slack.max())
            checked += 1
    return checked


@pytest.mark.parametrize("rows,frac", [(3_200_000, 0.05), (3_200_000, 0.2)])
def test_attn_window_scales_past_two_million_keys(dev, rows, frac):
    """The reference's top_k = 0.05 (and a larger fraction) on the c4 shape (128 numeric columns, MCAR 0.5) at
    3.2M rows = 1.6M valid keys per column and k = 160 000 / 640 000: with round 2's fixed 8 192-key sample the selected
    bin held ~ 5.5e-4 * nk keys and every pair past ~ 2M valid keys overflowed its 1 024-entry list into the six-pass radix
    fallback; the sample now grows with nk^2 (win_kstride, csrc/attn.hip).  A 256-row query shard against all keys:
    NO pair in the fallback, k-th logit and fills against the fp64 closed form."""
    ops = _ops()
    from ddgani_amd import synth
    k = int(rows * frac)
    tab = synth.make_table("c4", dev, rows=rows)
    lo, hi = rows // 2, rows // 2 + 256
    Xq, mq = tab.X_attn[lo:hi].contiguous(), tab.m8[lo:hi].contiguous()
    o, t, nv = ops.attn_topk_fill(Xq, mq, tab.X_attn, tab.m8, tab.W, tab.layout, k, q_row_offset=lo, debug=True, mode=1)
    torch.cuda.synchronize()
    n_fallback = ops.attn_fallback_pairs(hi - lo, rows, tab.layout, k, False, dev)
    fin = ~torch.isnan(t)
    assert int(fin.sum()) > 10_000 and int(nv[fin].min()) > 1_500_000
    assert n_fallback == 0, f"{n_fallback} of {int(fin.sum())} pairs took the radix fallback"
    assert _fp64_spot_pairs(tab, lo, mq.cpu().numpy(), o, t, k, 3, 2, seed=5) >= 6


def test_attn_large_k_c3_full_size(dev):
    """BASELINE configuration c3 at its full size under the reference's own top_k (param.json: 0.05 -> k = 50 000 of
    1M rows): a shard of 256 query rows against all keys.  Window selection (f16 split arithmetic) against the fp32
    radix path -- the k-th logit within 2e-7, fills within 1e-5 apart from flipped near-ties -- and spot pairs against an fp64 closed form with the
    near-tie slack of test_attn_large_k_at_scale."""
    ops = _ops()
    from ddgani_amd import synth
    rows, k = 1_000_000, 50_000
    tab = synth.make_table("c3", dev, rows=rows)
    lo, hi = 500_000, 500_256
    Xq, mq = tab.X_attn[lo:hi].contiguous(), tab.m8[lo:hi].contiguous()
    res = {}
    for radix in ("1", "0"):
        os.environ["DDG_ATTN_RADIX"] = radix
        try:
            o, t, nv = ops.attn_topk_fill(Xq, mq, tab.X_attn, tab.m8, tab.W, tab.layout, k, q_row_offset=lo, debug=True,
                                          mode=1)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("DDG_ATTN_RADIX", None)
        res[radix] = (o, t, nv)
    (o1, t1, n1), (o0, t0, n0) = res["1"], res["0"]
    fin = ~torch.isnan(t1)
    assert torch.equal(n1, n0) and torch.equal(torch.isnan(t0), torch.isnan(t1)) and int(fin.sum()) > 2000
    assert float((t0[fin] - t1[fin]).abs().max()) <= 2e-7
    # a k-th logit closer than the two arithmetics' difference to its neighbour moves ONE key of k in or out of the kept
    # set (exempt: within 1e-6 of the k-th logit): such a cell shifts by at most (value spread) / k
    d = (o0 - o1).abs()
    assert float((d > 1e-5).float().mean()) < 0.01
    assert float(d.max()) <= 1e-5 + 8.0 / k
    X64 = tab.X_attn.double()
    W64 = tab.W.double()
    m = tab.m8
    off = tab.layout.offsets
    mq_h = mq.cpu().numpy()
    rng = np.random.default_rng(11)
    with_missing = np.where((mq_h == 0).any(1))[0]
    checked = 0
    for j in rng.choice(with_missing, 4, replace=False):
        for c in np.where(mq_h[j] == 0)[0][:2]:
            keys = torch.nonzero(m[:, c] == 1).flatten()
            keys = keys[keys != lo + j]
            s = (X64[keys] @ (W64[c] * X64[lo + j])).cpu().numpy()
            kth = np.partition(s, len(s) - k)[len(s) - k]
            xv = X64[keys][:, off[c]:off[c + 1]].cpu().numpy()
            keep = s >= kth
            w = np.exp(s - s.max())
            Z = w[keep].sum()
            ref = (w[keep] @ xv[keep]) / Z
            band = np.abs(s - kth) <= 1e-6
            slack = (w[band, None] * np.abs(xv[band] - ref)).sum(0) / Z + 1e-5
            assert abs(float(t0[j, c]) - kth) <= 1e-6, (j, c, float(t0[j, c]), kth)
            got = o0[j, off[c]:off[c + 1]].double().cpu().numpy()
            assert np.all(np.abs(got - ref) <= slack), (j, c, np.abs(got - ref).max(), slack.max())
            checked += 1
    assert checked >= 6
