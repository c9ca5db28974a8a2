#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite
never execute this file, they only read the committed ``*.npz`` (plain arrays,
``allow_pickle=False``).  Nothing of the reference's source is copied: the reference is
imported from where it lies and called through its own public functions; only inputs,
the explicit random draws and the outputs are stored.

Import recipe = SURVEY.md appendix B: sys.path gets the reference's PyCharm source roots;
``torchmetrics`` (absent in this image) is only used by the reference for a running-mean
loss logger whose value is discarded (model/Diffusion.py:247,281,317), so a 6-line
MeanMetric accumulator is registered in sys.modules before ``model.Diffusion`` is imported.
It takes part in no arithmetic that reaches a fixture.

Every fixture is cross-checked here against oracle/ddgani_oracle.py before it is written, so
a fixture on disk is at the same time the pin of the oracle.
"""
import os
import random
import sys
import types

import numpy as np
import pandas as pd
import torch

R = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (R, R + "/utils", R + "/BaseLine", R + "/param"):
    sys.path.insert(0, p)
sys.path.insert(0, ROOT)

tm = types.ModuleType("torchmetrics")


class MeanMetric:  # logging accumulator only
    def __init__(self):
        self.s, self.n = 0.0, 0

    def update(self, v):
        self.s += float(v)
        self.n += 1

    def compute(self):
        return torch.tensor(self.s / max(self.n, 1))


tm.MeanMetric = MeanMetric
sys.modules["torchmetrics"] = tm

import util as ref_util                                  # noqa: E402  (reference utils/util.py)
from model import Diffusion as ref_diff                  # noqa: E402
from model import Discriminator_model as ref_disc        # noqa: E402
import field as ref_field                                # noqa: E402

from oracle import ddgani_oracle as orc                  # noqa: E402


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, {k: getattr(v, "shape", None) for k, v in arrs.items()})


# ------------------------------------------------------------------ synthetic tables
def make_table(rng, N, col_kinds, miss_rate):
    """col_kinds: list of 0 (numeric) or cardinality>0 (categorical).  Returns the table the
    imputer entry would hand to init_attn_2: numerics mean-prefilled (Mean.py semantics: sum of
    observed / N), categoricals 'Null' where missing, plus data_m."""
    C = len(col_kinds)
    data_m = (rng.uniform(size=(N, C)) < 1 - miss_rate).astype(np.int64)
    cols = {}
    latent = rng.normal(size=(N, 3))
    for c, kind in enumerate(col_kinds):
        if kind == 0:
            v = latent @ rng.normal(size=3) + 0.5 * rng.normal(size=N)
            v = np.round(v * 10.0 + 3.0 * c, 3)
            obs = data_m[:, c] == 1
            mean = v[obs].sum() / N
            v = np.where(obs, v, mean)
            cols[f"a{c}"] = v
        else:
            code = np.floor((latent @ rng.normal(size=3) * 0.7 + rng.normal(size=N) * 0.6) % kind).astype(int)
            v = np.array([f"v{z}" for z in code], dtype=object)
            v[data_m[:, c] == 0] = "Null"
            cols[f"a{c}"] = v
    return pd.DataFrame(cols), data_m


def corr_like(rng, C):
    w = np.abs(rng.uniform(size=(C, C))) * 1e-3
    w = (w + w.T) / 2
    np.fill_diagonal(w, 0.0)
    return w


def gen_attn(name, seed, N, col_kinds, miss_rate, top_k, corr_scale=1.0):
    rng = np.random.default_rng(seed)
    df, data_m = make_table(rng, N, col_kinds, miss_rate)
    C = len(col_kinds)
    cat_cols = [c for c, kd in enumerate(col_kinds) if kd > 0]
    con_cols = [c for c in range(C) if c not in cat_cols]
    value_cat = [f"a{c}" for c in cat_cols]
    corr = corr_like(rng, C) * corr_scale
    # encoder fitted on the complete vocabulary incl. "Null" so transform never fails
    enc = None
    if cat_cols:
        from sklearn.preprocessing import OrdinalEncoder
        enc = OrdinalEncoder()
        full = df[value_cat].copy()
        enc.fit(full)
    out_df, new_code = ref_util.init_attn_2(corr.copy(), df.copy(), data_m.copy(), cat_cols, enc, value_cat,
                                            torch.device("cpu"), top_k)
    new_code = new_code.numpy()

    # intermediates, obtained with the reference's own helpers (utils/util.py:398-407)
    code = df.copy()
    for c in cat_cols:
        vc = code[f"a{c}"].value_counts()
        for val in vc.index.tolist():
            if val != "Null":
                code[f"a{c}"] = code[f"a{c}"].apply(lambda x: val if x == "Null" else x)
    code, _ = ref_util.categorical_to_code(code, value_cat, enc)
    code.columns = list(range(C))
    fields, enc_code = ref_util.Data_convert(code, "mean_std", con_cols)
    X = enc_code.values.astype(np.float32)
    offsets = [0]
    for f in fields:
        offsets.append(offsets[-1] + f.dim())
    W = orc.expand_cols(corr, offsets).astype(np.float32)
    k = int(N * top_k)

    # pin the oracle: same op sequence must reproduce the reference to the last bit
    o = orc.attn_fill_dense(torch.tensor(X), torch.tensor(W), data_m, offsets, k).numpy()
    diff = np.abs(o - new_code).max()
    print(f"[{name}] oracle dense vs reference max abs diff = {diff:.3e}")
    assert diff == 0.0, diff
    cf, _ = orc.attn_fill_closed_form(torch.tensor(X), torch.tensor(W), data_m, offsets, k)
    print(f"[{name}] closed form vs reference max abs diff = {np.abs(cf - new_code).max():.3e}")

    num_values = np.full((N, C), np.nan)
    cat_values = np.full((N, C), "", dtype="<U16")
    for c in range(C):
        if c in cat_cols:
            cat_values[:, c] = df[f"a{c}"].values.astype(str)
        else:
            num_values[:, c] = df[f"a{c}"].values.astype(float)
    out_num = np.full((N, C), np.nan)
    out_cat = np.full((N, C), "", dtype="<U16")
    for c in range(C):
        if c in cat_cols:
            out_cat[:, c] = out_df[f"a{c}"].values.astype(str)
        else:
            out_num[:, c] = out_df[f"a{c}"].values.astype(float)
    save(name, num_values=num_values, cat_values=cat_values, data_m=data_m, corr=corr,
         categorical_cols=np.array(cat_cols, dtype=np.int64), top_k=np.array(top_k), k=np.array(k),
         X=X, W=W, offsets=np.array(offsets, dtype=np.int64), new_code=new_code,
         out_num=out_num, out_cat=out_cat)


# ------------------------------------------------------------------ network fixtures
class _F:  # field stand-in with the two attributes Generator_x0 reads
    def __init__(self, cat, d):
        self.data_type = "Categorical Data" if cat else "Numerical Data"
        self._d = d

    def dim(self):
        return self._d


def build_nets(widths, T, seed=3407):
    """Construction order of BaseLine/DiffGANI.py:47-54 under torch.manual_seed(3407)."""
    fields = [_F(w > 1, w) for w in widths]
    D = sum(widths)
    C = len(widths)
    torch.manual_seed(seed)
    ref_diff.Diffusion(D, D, D, T + 1)
    ref_disc.D(D, D // 2, C)
    Dn = ref_disc.Discriminator_noise_x(D * 2, D // 2, C, T + 1)
    G = ref_diff.Generator_x0(D, D * 2, D, T + 1, fields)
    return G, Dn, fields


def sd_np(mod, prefix):
    return {prefix + k.replace(".", "__"): v.detach().numpy().copy() for k, v in mod.state_dict().items()}


def sd_t(mod):
    return {k: v.detach().clone() for k, v in mod.state_dict().items()}


def layout(widths):
    offsets = [0]
    for w in widths:
        offsets.append(offsets[-1] + w)
    return offsets, [w > 1 for w in widths]


def synth_code(rng, N, widths, miss_rate):
    cols = []
    for w in widths:
        if w == 1:
            cols.append(rng.uniform(size=(N, 1)))
        else:
            oh = np.zeros((N, w))
            oh[np.arange(N), rng.integers(0, w, size=N)] = 1
            cols.append(oh)
    x = np.concatenate(cols, axis=1).astype(np.float32)
    data_m = (rng.uniform(size=(N, len(widths))) < 1 - miss_rate).astype(np.float32)
    offsets, _ = layout(widths)
    M = orc.expand_cols(data_m, offsets).astype(np.float32)
    return x, data_m, M


def gen_sampler(name, seed, N, widths, T):
    rng = np.random.default_rng(seed)
    G, Dn, fields = build_nets(widths, T)
    G.train()
    Dn.train()
    # perturb BN affine so it is not the identity
    with torch.no_grad():
        for i in (1, 5, 9):
            G.linears[i].weight.add_(torch.tensor(rng.normal(size=G.linears[i].weight.shape) * 0.1, dtype=torch.float32))
            G.linears[i].bias.add_(torch.tensor(rng.normal(size=G.linears[i].bias.shape) * 0.1, dtype=torch.float32))
    x_np, data_m, M_np = synth_code(rng, N, widths, 0.3)
    x = torch.tensor(x_np)
    M = torch.tensor(M_np)
    D = x.shape[1]
    offsets, is_cat = layout(widths)
    DS = ref_diff.Diffusion_setting(T, D, torch.device("cpu"))
    PG, PD = sd_t(G), sd_t(Dn)
    S = orc.schedule(T)
    for kname in ("betas", "alphas", "sqrt_alphas", "alpha_cumulative", "sqrt_alpha_cumulative",
                  "sqrt_one_minus_alpha_cumulative"):
        assert torch.equal(S[kname], getattr(DS, kname)), kname

    # ---- reference run
    torch.manual_seed(1234 + seed)
    with torch.no_grad():
        x0_hat, d_out = ref_diff.reverse_diffusion_G_D_sample(G, Dn, DS, T, N, D, torch.device("cpu"), x, M)

    # ---- replay the RNG stream to make every draw explicit
    torch.manual_seed(1234 + seed)
    torch.randint(low=T, high=T + 1, size=(N,))
    eps_T = torch.randn_like(x)
    zs, masks = [], []

    def draw_masks():
        return [torch.empty(N, w).bernoulli_(0.8).div_(0.8) for w in (2 * D, D, 2 * D)]
    for ts in reversed(range(2, T + 1)):
        if ts > 2:
            zs.append(torch.randn_like(x))
            masks.append(draw_masks())
        else:
            masks.append(draw_masks())
            eps2 = torch.randn_like(x)
    with torch.no_grad():
        o = orc.reverse_sample(PG, S, T, x, M, offsets, is_cat, eps_T, zs, masks)
        od = orc.sampler_disc_tail(PD, S, x, o, eps2)
    d1 = (o - x0_hat).abs().max().item()
    d2 = (od - d_out).abs().max().item()
    print(f"[{name}] oracle sampler vs reference: x0_hat {d1:.3e}  disc {d2:.3e}")
    assert d1 < 2e-6 and d2 < 2e-6, (d1, d2)

    # single pieces taken from the same run: generator forward at t=T and one reverse step
    with torch.no_grad():
        tT = torch.full((N,), T, dtype=torch.long)
        x_T = orc.forward_noise(S, x, tT, eps_T)
        pred_T, pre_T = orc.generator_forward(PG, x_T, tT, offsets, is_cat, masks[0], return_pre=True)
        x_Tm1 = orc.reverse_step(S, x_T, x, M, pred_T, zs[0], tT)
        code = orc.mask_merge(x, M, x0_hat, offsets, is_cat)
        # reference generator called directly with the replayed masks' seed
        torch.manual_seed(77)
        ref_pred = G(x_T, tT)
        torch.manual_seed(77)
        mm = draw_masks()
        my_pred = orc.generator_forward(PG, x_T, tT, offsets, is_cat, mm)
        d3 = (ref_pred - my_pred).abs().max().item()
        print(f"[{name}] oracle generator fwd vs reference: {d3:.3e}")
        assert d3 < 1e-6
        # reference reverse step
        torch.manual_seed(78)
        ref_step = ref_diff.G_D_onestep(G, DS, x_T, tT, M, x)
        torch.manual_seed(78)
        zz = torch.randn_like(x)
        mm2 = draw_masks()
        my_step = orc.reverse_step(S, x_T, x, M, orc.generator_forward(PG, x_T, tT, offsets, is_cat, mm2), zz, tT)
        d4 = (ref_step - my_step).abs().max().item()
        print(f"[{name}] oracle reverse step vs reference: {d4:.3e}")
        assert d4 < 1e-6

    arrs = dict(widths=np.array(widths, dtype=np.int64), T=np.array(T), x=x_np, M=M_np, data_m=data_m,
                eps_T=eps_T.numpy(), eps2=eps2.numpy(), x0_hat=x0_hat.numpy(), d_out=d_out.numpy(),
                x_T=x_T.numpy(), pred_T=pred_T.numpy(), pre_T=pre_T.numpy(), x_Tm1=x_Tm1.numpy(),
                code=code.numpy())
    for i, z in enumerate(zs):
        arrs[f"z{i}"] = z.numpy()
    for i, ms in enumerate(masks):
        for l, m_ in enumerate(ms):
            arrs[f"mask{i}_{l}"] = (m_.numpy() != 0)      # stored as bool, value = 1/0.8 where True
    arrs.update(sd_np(G, "G__"))
    arrs.update(sd_np(Dn, "D__"))
    save(name, **arrs)


def gen_train(name, seed, N, widths, T):
    """Losses (F6) and one Adam step of each net (F7) on a B=128 batch, explicit randomness."""
    rng = np.random.default_rng(seed)
    G, Dn, fields = build_nets(widths, T)
    G.train()
    Dn.train()
    x_np, data_m, M_np = synth_code(rng, N, widths, 0.3)
    x = torch.tensor(x_np)
    M = torch.tensor(M_np)
    m = torch.tensor(data_m)
    D = x.shape[1]
    offsets, is_cat = layout(widths)
    DS = ref_diff.Diffusion_setting(T, D, torch.device("cpu"))
    S = orc.schedule(T)
    B = 128
    idx = rng.choice(N, B, replace=False)
    x0b, Mb, mb = x[idx], M[idx], m[idx]
    torch.manual_seed(4321 + seed)
    t = torch.randint(low=2, high=T + 1, size=(B,))
    eps = torch.randn_like(x0b)
    masks = [torch.empty(B, w).bernoulli_(0.8).div_(0.8) for w in (2 * D, D, 2 * D)]
    PG0, PD0 = sd_t(G), sd_t(Dn)
    w_res, w_gd, w_d = 100.0, 1.0, 0.2

    # ---- reference generator step (model/Diffusion.py:287-311 without the inert FD/L constants)
    optG = torch.optim.Adam(G.parameters(), lr=0.0001, betas=(0.9, 0.999), eps=1e-08, weight_decay=0.0001)
    optD = torch.optim.Adam(Dn.parameters(), lr=0.0001)
    torch.manual_seed(4321 + seed)
    t_r = torch.randint(low=2, high=T + 1, size=(B,))
    x_t, eps_r = ref_diff.forward_diffusion(DS, x0b, t_r)
    assert torch.equal(t, t_r) and torch.equal(eps, eps_r)
    pred = G(x_t, t_r)
    Lres = ref_diff.res_loss_func(fields, pred, x0b, Mb)
    Lgd = ref_diff.G_d_loss_func(DS, pred, x_t, x0b, t_r, Dn, Mb, eps_r, mb)
    lossG = Lres * w_res + Lgd * w_gd
    optG.zero_grad()
    lossG.backward()
    gradsG = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in G.named_parameters()}
    optG.step()
    PG1 = sd_t(G)

    # oracle check of forward + losses
    with torch.no_grad():
        my_xt = orc.forward_noise(S, x0b, t, eps)
        my_pred = orc.generator_forward(PG0, my_xt, t, offsets, is_cat, masks)
        d_pred = (my_pred - pred).abs().max().item()
        my_res = orc.res_loss(my_pred, x0b, Mb, offsets, is_cat)
        my_gd = orc.g_d_loss(PD0, S, my_pred, my_xt, x0b, t, Mb, eps, mb)
    print(f"[{name}] pred {d_pred:.3e} res {abs(my_res.item()-Lres.item()):.3e} gd {abs(my_gd.item()-Lgd.item()):.3e}")
    assert d_pred < 1e-6 and abs(my_res.item() - Lres.item()) < 1e-6 and abs(my_gd.item() - Lgd.item()) < 1e-6

    # ---- reference discriminator step (model/Diffusion.py:356-372; GradScaler is a no-op on CPU)
    # uses the UPDATED generator, fresh randomness
    rs = torch.get_rng_state()
    t2 = torch.randint(low=2, high=T + 1, size=(B,))
    x_t2, eps2 = ref_diff.forward_diffusion(DS, x0b, t2)
    pred2 = G(x_t2, t2)
    Ld = ref_diff.D_loss_func(DS, pred2, x_t2, x0b, t2, Dn, Mb, eps2, mb)
    optD.zero_grad(set_to_none=True)
    Ld.backward()
    gradsD = {k: p.grad.detach().clone() for k, p in Dn.named_parameters()}
    optD.step()
    PD1 = sd_t(Dn)
    torch.set_rng_state(rs)
    torch.randint(low=2, high=T + 1, size=(B,))
    torch.randn_like(x0b)
    masks2 = [torch.empty(B, w).bernoulli_(0.8).div_(0.8) for w in (2 * D, D, 2 * D)]
    with torch.no_grad():
        my_xt2 = orc.forward_noise(S, x0b, t2, eps2)
        my_pred2 = orc.generator_forward(PG1, my_xt2, t2, offsets, is_cat, masks2)
        my_d = orc.d_loss(PD0, S, my_pred2, my_xt2, x0b, t2, Mb, eps2, mb)
    print(f"[{name}] d-step pred {(my_pred2-pred2).abs().max().item():.3e} dloss {abs(my_d.item()-Ld.item()):.3e}")
    assert abs(my_d.item() - Ld.item()) < 1e-6

    arrs = dict(widths=np.array(widths, dtype=np.int64), T=np.array(T), x0=x0b.numpy(), M=Mb.numpy(), m=mb.numpy(),
                t=t.numpy(), eps=eps.numpy(), t2=t2.numpy(), eps2=eps2.numpy(),
                pred=pred.detach().numpy(), pred2=pred2.detach().numpy(),
                res_loss=np.array(Lres.item()), gd_loss=np.array(Lgd.item()), d_loss=np.array(Ld.item()),
                w_res=np.array(w_res), w_gd=np.array(w_gd), w_d=np.array(w_d))
    for l in range(3):
        arrs[f"mask_{l}"] = masks[l].numpy() != 0
        arrs[f"mask2_{l}"] = masks2[l].numpy() != 0
    for k, v in PG0.items():
        arrs["G0__" + k.replace(".", "__")] = v.numpy()
    for k, v in PG1.items():
        arrs["G1__" + k.replace(".", "__")] = v.numpy()
    for k, v in PD0.items():
        arrs["D0__" + k.replace(".", "__")] = v.numpy()
    for k, v in PD1.items():
        arrs["D1__" + k.replace(".", "__")] = v.numpy()
    for k, v in gradsG.items():
        if v is not None:
            arrs["gG__" + k.replace(".", "__")] = v.numpy()
    for k, v in gradsD.items():
        arrs["gD__" + k.replace(".", "__")] = v.numpy()
    save(name, **arrs)


def gen_schedule():
    arrs = {}
    for T in (6, 10):
        DS = ref_diff.Diffusion_setting(T, 4, torch.device("cpu"))
        for kname in ("betas", "alphas", "sqrt_alphas", "one_by_sqrt_alpha_s", "sqrt_beta_s", "alpha_cumulative",
                      "sqrt_alpha_cumulative", "sqrt_one_minus_alpha_cumulative"):
            arrs[f"T{T}_{kname}"] = getattr(DS, kname).numpy()
            assert torch.equal(orc.schedule(T)[kname], getattr(DS, kname))
    save("schedule", **arrs)


def gen_error_loss(seed=5):
    rng = np.random.default_rng(seed)
    N, C = 300, 6
    con = [0, 2, 3, 5]
    cat = [1, 4]
    ori = {}
    for c in range(C):
        ori[f"a{c}"] = (rng.normal(size=N) * (c + 1) + c) if c in con else np.array([f"v{z}" for z in rng.integers(0, 4, N)], dtype=object)
    ori = pd.DataFrame(ori)
    data_m = (rng.uniform(size=(N, C)) < 0.7).astype(np.int64)
    data_m[:, 5] = 1          # a column without missing cells is skipped from the average
    imp = ori.copy()
    for c in range(C):
        miss = data_m[:, c] == 0
        if c in con:
            imp.loc[miss, f"a{c}"] = ori.loc[miss, f"a{c}"] + rng.normal(size=miss.sum())
        else:
            imp.loc[miss, f"a{c}"] = np.array([f"v{z}" for z in rng.integers(0, 4, miss.sum())], dtype=object)
    value_cat = [f"a{c}" for c in cat]
    from sklearn.preprocessing import OrdinalEncoder
    enc = OrdinalEncoder().fit(ori[value_cat])
    armse, amae = ref_util.errorLoss(imp, ori, data_m, value_cat, con, enc)
    oc, ic = ori.copy(), imp.copy()
    oc[value_cat] = enc.transform(oc[value_cat])
    ic[value_cat] = enc.transform(ic[value_cat])
    a2, m2 = orc.error_loss_coded(ic.values.astype(float), oc.values.astype(float), data_m, con)
    print(f"[error_loss] ref {armse:.10f} {amae:.10f}  oracle {a2:.10f} {m2:.10f}")
    assert abs(a2 - armse) < 1e-12 and abs(m2 - amae) < 1e-12
    save("error_loss", imputed=ic.values.astype(float), ori=oc.values.astype(float), data_m=data_m,
         continuous_cols=np.array(con, dtype=np.int64), armse=np.array(armse), amae=np.array(amae))


if __name__ == "__main__":
    which = sys.argv[1:] or ["schedule", "attn", "sampler", "train", "error"]
    np.random.seed(0)
    random.seed(0)
    if "schedule" in which:
        gen_schedule()
    if "attn" in which:
        gen_attn("attn_num64", 0, 64, [0] * 8, 0.3, 0.2)
        gen_attn("attn_mixed257", 1, 257, [0, 0, 5, 0, 3, 0, 0, 7, 0, 0], 0.3, 0.05)
        gen_attn("attn_onehot200", 2, 200, [4, 6, 3, 0, 5, 8, 2], 0.2, 0.05)
        gen_attn("attn_num600", 3, 600, [0] * 12, 0.5, 0.05)
    if "sampler" in which:
        gen_sampler("sampler_num", 0, 257, [1] * 13, 6)
        gen_sampler("sampler_mixed", 1, 300, [1, 1, 5, 1, 3, 1, 1, 7, 1, 1], 6)
        gen_sampler("sampler_T10", 2, 130, [1, 4, 1, 1, 6, 1], 10)
    if "train" in which:
        gen_train("train_mixed", 0, 400, [1, 1, 5, 1, 3, 1, 1, 7, 1, 1], 6)
        gen_train("train_num", 1, 300, [1] * 13, 6)
    if "error" in which:
        gen_error_loss()
