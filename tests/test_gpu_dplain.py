"""GPU: the plain discriminator that get_valid_data_index trains when fewer than 20 % of the rows are complete
(reference utils/util.py:623-646, model/Discriminator_model.py:5-39) through the C-ABI: ddg_dplain_grads / ddg_dplain_fwd
against torch autograd on the same weights, inputs and dropout keeps -- at a few hundred rows (the 32 x 32-tile layer
kernel, one row chunk of the weight-gradient reduction) and at 20 000 rows (the wide f16-split layer kernel, five row
chunks); one Adam step against torch.optim.Adam.  The selected validation rows on the reference's `hospital` run are
checked inside tests/test_gpu_e2e.py (they must be the reference's)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(sd, x, m, keeps):
    W = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
    h1 = torch.relu(x.double() @ W["fc1.weight"].T + W["fc1.bias"]) * keeps[0].double() / 0.8
    h2 = torch.relu(h1 @ W["fc2.weight"].T + W["fc2.bias"]) * keeps[1].double() / 0.8
    p = torch.sigmoid(h2 @ W["fc3.weight"].T + W["fc3.bias"])
    loss = -torch.mean(m.double() * torch.log(p + 1e-8) + (1 - m.double()) * torch.log(1 - p + 1e-8))
    loss.backward()
    return p.detach(), {k: v.grad for k, v in W.items()}


@pytest.mark.parametrize("rows,D,C", [(700, 37, 9), (333, 145, 9), (20_000, 64, 12)])
def test_dplain_grads_and_forward_match_autograd(rows, D, C):
    from ddgani_amd import ops
    from ddgani_amd.model.Discriminator_model import D as Dplain
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    net = Dplain(D, D // 2, C)
    x = torch.rand(rows, D)
    m = (torch.rand(rows, C) < 0.8).float()
    keeps = [torch.empty(rows, D).bernoulli_(0.8).to(torch.uint8) for _ in range(2)]
    sd = {k: v.detach().clone() for k, v in net.state_dict().items() if k in ops.DPLAIN_KEYS}
    p_ref, g_ref = _reference(sd, x, m, keeps)
    P = ops.DplainParams(sd, dev)
    buf = ops.ParamGrads(P, ops.DPLAIN_KEYS, n_extra=0)
    kd = [k.to(dev) for k in keeps]
    ops.dplain_grads(P, buf, x.to(dev), m.to(dev), drops=kd)
    p = ops.dplain_fwd(P, x.to(dev), drops=kd)
    torch.cuda.synchronize()
    assert float((p.double().cpu() - p_ref).abs().max()) < 2e-6
    for k in ops.DPLAIN_KEYS:
        got, ref = buf.g[k].double().cpu(), g_ref[k]
        rel = float((got - ref).abs().max() / ref.abs().max())
        assert rel < 1e-4, (k, rel)
    # one Adam step == torch.optim.Adam(lr 0.002) on the same gradients
    params = {k: sd[k].clone().requires_grad_(True) for k in ops.DPLAIN_KEYS}
    for k in ops.DPLAIN_KEYS:
        params[k].grad = buf.g[k].cpu().clone()
    opt = torch.optim.Adam(list(params.values()), lr=0.002)
    opt.step()
    buf.adam_step(lr=0.002)
    for k in ops.DPLAIN_KEYS:
        assert float((P.t[k].cpu() - params[k].detach()).abs().max()) < 2e-7, k


def test_dplain_on_chip_dropout_keeps_four_fifths():
    """drops=None: Philox keeps on the device -- about 80 % of the units of a layer survive and two calls with different
    stream numbers differ"""
    from ddgani_amd import ops
    from ddgani_amd.model.Discriminator_model import D as Dplain
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    net = Dplain(32, 16, 5)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items() if k in ops.DPLAIN_KEYS}
    P = ops.DplainParams(sd, dev)
    x = torch.rand(4096, 32, device=dev)
    a = ops.dplain_fwd(P, x, seed=3, stream_id=1)
    b = ops.dplain_fwd(P, x, seed=3, stream_id=2)
    c = ops.dplain_fwd(P, x, seed=3, stream_id=1)
    assert torch.equal(a, c) and not torch.equal(a, b)
    assert bool(torch.isfinite(a).all()) and float(a.min()) > 0 and float(a.max()) < 1
