"""CPU: the oracle (oracle/ddgani_oracle.py) against the golden vectors made by running the
reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import ddgani_oracle as orc
from util_golden import layout_of, load, sampler_randomness, state_dict


@pytest.mark.parametrize("T", [6, 10])
def test_schedule(T):
    z = load("schedule")
    S = orc.schedule(T)
    for k, v in S.items():
        assert np.array_equal(v.numpy(), z[f"T{T}_{k}"]), k


@pytest.mark.parametrize("name", ["attn_num64", "attn_mixed257", "attn_onehot200", "attn_num600", "attn_adult2000"])
def test_attn_dense_bit_exact(name):
    z = load(name)
    out = orc.attn_fill_dense(torch.tensor(z["X"]), torch.tensor(z["W"]), z["data_m"], z["offsets"].tolist(),
                              int(z["k"])).numpy()
    assert np.array_equal(out, z["new_code"])


@pytest.mark.parametrize("name", ["attn_num64", "attn_mixed257", "attn_onehot200", "attn_num600", "attn_adult2000"])
def test_attn_closed_form_near_tie_aware(name):
    """closed form == reference except on cells flagged as near-ties (another logit within 1e-6 of
    the k-th), where a membership flip may move the value by up to ~1/k of the key spread."""
    z = load(name)
    off = z["offsets"].tolist()
    cf, info = orc.attn_fill_closed_form(torch.tensor(z["X"]), torch.tensor(z["W"]), z["data_m"], off, int(z["k"]),
                                         tie_tol=1e-6)
    ref = z["new_code"]
    nflag = 0
    for (j, c), d in info.items():
        a, b = off[c], off[c + 1]
        err = np.abs(cf[j, a:b] - ref[j, a:b]).max()
        if d["near_tie"]:
            nflag += 1
            assert err < 1.0
        else:
            assert err < 2e-6, (j, c, err)
    assert nflag < 0.1 * len(info)
    obs = np.repeat(z["data_m"], np.diff(off), axis=1) == 1
    assert np.abs(cf - ref)[obs].max() < 1e-7


@pytest.mark.parametrize("name,T", [("sampler_num", 6), ("sampler_mixed", 6), ("sampler_T10", 10)])
def test_sampler(name, T):
    z = load(name)
    off, cat = layout_of(z["widths"])
    PG, PD = state_dict(z, "G__"), state_dict(z, "D__")
    S = orc.schedule(T)
    x, M = torch.tensor(z["x"]), torch.tensor(z["M"])
    zs, _, masks = sampler_randomness(z, T)
    with torch.no_grad():
        out = orc.reverse_sample(PG, S, T, x, M, off, cat, torch.tensor(z["eps_T"]), zs, masks)
        d = orc.sampler_disc_tail(PD, S, x, out, torch.tensor(z["eps2"]))
        code = orc.mask_merge(x, M, out, off, cat)
    assert np.array_equal(out.numpy(), z["x0_hat"])
    assert np.array_equal(d.numpy(), z["d_out"])
    assert np.array_equal(code.numpy(), z["code"])


@pytest.mark.parametrize("name", ["train_mixed", "train_num"])
def test_losses(name):
    z = load(name)
    off, cat = layout_of(z["widths"])
    T = int(z["T"])
    S = orc.schedule(T)
    PG0, PG1, PD0 = state_dict(z, "G0__"), state_dict(z, "G1__"), state_dict(z, "D0__")
    x0, M, m = torch.tensor(z["x0"]), torch.tensor(z["M"]), torch.tensor(z["m"])
    t, eps = torch.tensor(z["t"]), torch.tensor(z["eps"])
    masks = [torch.tensor(z[f"mask_{l}"].astype(np.float32) / np.float32(0.8)) for l in range(3)]
    with torch.no_grad():
        xt = orc.forward_noise(S, x0, t, eps)
        pred = orc.generator_forward(PG0, xt, t, off, cat, masks)
        assert np.array_equal(pred.numpy(), z["pred"])
        assert abs(orc.res_loss(pred, x0, M, off, cat).item() - float(z["res_loss"])) < 1e-7
        assert abs(orc.g_d_loss(PD0, S, pred, xt, x0, t, M, eps, m).item() - float(z["gd_loss"])) < 1e-7
        t2, eps2 = torch.tensor(z["t2"]), torch.tensor(z["eps2"])
        masks2 = [torch.tensor(z[f"mask2_{l}"].astype(np.float32) / np.float32(0.8)) for l in range(3)]
        xt2 = orc.forward_noise(S, x0, t2, eps2)
        pred2 = orc.generator_forward(PG1, xt2, t2, off, cat, masks2)
        assert abs(orc.d_loss(PD0, S, pred2, xt2, x0, t2, M, eps2, m).item() - float(z["d_loss"])) < 1e-7


@pytest.mark.parametrize("name", ["train_mixed", "train_num"])
def test_adam_step(name):
    """weights after one torch.optim.Adam step of the reference == oracle.adam_step on its grads."""
    z = load(name)
    for pre, g0, g1, wd in (("gG__", "G0__", "G1__", 1e-4), ("gD__", "D0__", "D1__", 0.0)):
        for k in z.files:
            if not k.startswith(pre):
                continue
            nm = k[len(pre):]
            p0, p1, g = torch.tensor(z[g0 + nm]), torch.tensor(z[g1 + nm]), torch.tensor(z[k])
            p, _, _ = orc.adam_step(p0, g, torch.zeros_like(p0), torch.zeros_like(p0), 1, wd=wd)
            assert torch.allclose(p, p1, rtol=0, atol=2e-7), nm


def test_error_loss():
    z = load("error_loss")
    a, m = orc.error_loss_coded(z["imputed"], z["ori"], z["data_m"], z["continuous_cols"].tolist())
    assert abs(a - float(z["armse"])) < 1e-12 and abs(m - float(z["amae"])) < 1e-12


@pytest.mark.parametrize("name", ["learner_small", "learner_wide", "learner_manyclass"])
def test_learner_oracle_matches_reference(name):
    """8(f) row 1: the oracle's explicit-gradient restatement of train_L_code against the reference's outputs."""
    z = load(name)
    T = torch.tensor
    loss, acc, W, b = orc.learner_train(int(z["epoch"]), T(z["x_train"]), T(z["y_train"]), T(z["x_valid"]),
                                        T(z["y_valid"]), int(z["L"]))
    assert abs(loss - float(z["val_loss"])) < 2e-6
    assert acc == float(z["acc"])
    np.testing.assert_allclose(W.numpy(), z["W"], rtol=0, atol=1e-6)
