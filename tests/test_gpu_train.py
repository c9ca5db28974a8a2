"""GPU: the training step (ddg_train_gen_grads / ddg_train_disc_grads / ddg_adam_step) through the
C-ABI against the golden vectors of the reference's own generator / discriminator step
(tests/golden/train_*.npz: losses, autograd gradients, weights after optimizer.step()).

Tolerances: losses 2e-6 abs; gradients 1e-4 relative to the largest entry of the tensor
(fp32 summation order differs from ATen's); weights after one Adam step within 2e-7 abs wherever the
update is not sign-ambiguous (|grad| well above fp32 noise), since lr = 1e-4 bounds the step size.
"""
import numpy as np
import pytest
import torch

from util_golden import load, state_dict

pytestmark = pytest.mark.gpu

G2ABI = {"linears.0.weight": "W1", "linears.0.bias": "b1", "linears.1.weight": "g1", "linears.1.bias": "be1",
         "linears.4.weight": "W2", "linears.4.bias": "b2", "linears.5.weight": "g2", "linears.5.bias": "be2",
         "linears.8.weight": "W3", "linears.8.bias": "b3", "linears.9.weight": "g3", "linears.9.bias": "be3",
         "linears.12.weight": "W4", "linears.12.bias": "b4", "time_embedding.weight": "temb"}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _setup(z, dev, gprefix="G0__", dprefix="D0__"):
    from ddgani_amd import ops
    T = int(z["T"])
    lay = ops.FieldLayout(z["widths"].tolist(), dev)
    PG = ops.GenParams(state_dict(z, gprefix), dev)
    PD = ops.DiscParams(state_dict(z, dprefix), dev)
    sched = torch.from_numpy(ops.schedule(T)).to(dev)
    up = lambda k, dt=None: torch.tensor(z[k] if dt is None else z[k].astype(dt)).to(dev)
    return ops, T, lay, PG, PD, sched, up


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


@pytest.mark.parametrize("name", ["train_mixed", "train_num"])
def test_generator_step(dev, name):
    z = load(name)
    ops, T, lay, PG, PD, sched, up = _setup(z, dev)
    grads = ops.ParamGrads(PG, ops.GEN_KEYS)
    drops = [up(f"mask_{l}", np.uint8) for l in range(3)]
    losses, pred = ops.train_gen_grads(PG, PD, grads, up("x0"), up("M"), up("m"), up("t"), lay, sched, T,
                                       float(z["w_res"]), float(z["w_gd"]), eps=up("eps"), drops=drops,
                                       want_pred=True)
    np.testing.assert_allclose(pred.cpu().numpy(), z["pred"], rtol=0, atol=1e-5)
    l = losses.cpu().numpy()
    assert abs(l[0] - float(z["res_loss"])) < 2e-6 and abs(l[1] - float(z["gd_loss"])) < 2e-6, l
    for k in ops.GEN_KEYS:
        ref = z["gG__" + k.replace(".", "__")]
        got = grads.g[k].cpu().numpy()
        if k.endswith("bias") and k in ("linears.0.bias", "linears.4.bias", "linears.8.bias"):
            # bias in front of a train-mode BatchNorm: the true gradient is 0, both sides hold rounding noise
            assert np.abs(got).max() < 1e-5 * max(1.0, np.abs(z["gG__linears__12__bias"]).max() * 100), k
            continue
        assert _rel(got, ref) < 1e-4, (k, _rel(got, ref))
    # Adam with coupled weight decay -> weights after the step
    grads.adam_step(lr=1e-4, weight_decay=1e-4)
    for k in ops.GEN_KEYS:
        ref = z["G1__" + k.replace(".", "__")]
        got = PG.t[k].cpu().numpy()
        before = z["G0__" + k.replace(".", "__")]
        g = z["gG__" + k.replace(".", "__")] + 1e-4 * before
        assert np.abs(got - ref).max() < 2.1e-4, k          # never more than two step sizes away
        if k in ("linears.0.bias", "linears.4.bias", "linears.8.bias"):
            continue       # gradient = rounding noise on both sides (see above): Adam normalises noise to +-lr
        stable = np.abs(g) > 1e-6 * max(np.abs(g).max(), 1e-30) + 1e-7
        assert np.abs(got - ref)[stable].max() < 2e-7, (k, np.abs(got - ref)[stable].max())


@pytest.mark.parametrize("name", ["train_mixed", "train_num"])
def test_discriminator_step(dev, name):
    z = load(name)
    # the reference's D step runs with the generator already updated (G1) and fresh randomness
    ops, T, lay, PG, PD, sched, up = _setup(z, dev, gprefix="G1__")
    grads = ops.ParamGrads(PD, ops.DISC_KEYS)
    drops = [up(f"mask2_{l}", np.uint8) for l in range(3)]
    loss, pred = ops.train_disc_grads(PG, PD, grads, up("x0"), up("M"), up("m"), up("t2"), lay, sched, T,
                                      loss_scale=1.0, eps=up("eps2"), drops=drops, want_pred=True)
    np.testing.assert_allclose(pred.cpu().numpy(), z["pred2"], rtol=0, atol=1e-5)
    assert abs(float(loss.item()) - float(z["d_loss"])) < 2e-6
    for k in ops.DISC_KEYS:
        ref = z["gD__" + k.replace(".", "__")]
        got = grads.g[k].cpu().numpy()
        assert _rel(got, ref) < 1e-4, (k, _rel(got, ref))
    grads.adam_step(lr=1e-4)
    for k in ops.DISC_KEYS:
        ref = z["D1__" + k.replace(".", "__")]
        got = PD.t[k].cpu().numpy()
        g = z["gD__" + k.replace(".", "__")]
        stable = np.abs(g) > 1e-6 * max(np.abs(g).max(), 1e-30) + 1e-9
        assert np.abs(got - ref)[stable].max() < 2e-7, (k, np.abs(got - ref)[stable].max())
        assert np.abs(got - ref).max() < 2.1e-4, k


def test_loss_scale_and_onchip_randomness(dev):
    """loss_scale multiplies the discriminator gradients (D_weight on epochs % 100 == 0); on-chip
    noise / dropout is reproducible per (seed, stream) and gives finite gradients."""
    z = load("train_mixed")
    ops, T, lay, PG, PD, sched, up = _setup(z, dev)
    ga, gb = ops.ParamGrads(PD, ops.DISC_KEYS), ops.ParamGrads(PD, ops.DISC_KEYS)
    args = (up("x0"), up("M"), up("m"), up("t"), lay, sched, T)
    ops.train_disc_grads(PG, PD, ga, *args, loss_scale=1.0, seed=3, stream_id=9)
    ops.train_disc_grads(PG, PD, gb, *args, loss_scale=0.2, seed=3, stream_id=9)
    for k in ops.DISC_KEYS:
        assert torch.isfinite(ga.g[k]).all()
        assert torch.allclose(ga.g[k] * 0.2, gb.g[k], rtol=1e-5, atol=1e-9), k
    gg = ops.ParamGrads(PG, ops.GEN_KEYS)
    l1, _ = ops.train_gen_grads(PG, PD, gg, *args, 100.0, 1.0, seed=3, stream_id=9)
    w1 = gg.g["linears.12.weight"].clone()
    l2, _ = ops.train_gen_grads(PG, PD, gg, *args, 100.0, 1.0, seed=3, stream_id=9)
    assert torch.equal(w1, gg.g["linears.12.weight"]) and torch.isfinite(l1).all()
    l3, _ = ops.train_gen_grads(PG, PD, gg, *args, 100.0, 1.0, seed=4, stream_id=9)
    assert not torch.equal(w1, gg.g["linears.12.weight"])
