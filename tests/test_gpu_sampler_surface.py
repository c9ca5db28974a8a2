"""GPU: the drop-in sampler surface of model/Diffusion.py (G_D_onestep, reverse_diffusion_G_D_sample, forward_diffusion,
get_xt) called the way a script written against the reference calls it, against the reference's own outputs
(tests/golden/sampler_*.npz, made by tests/golden/make_golden.py), and the freshness of the on-chip random draws when
the caller passes no randomness (the reference draws from torch's global generator on every call)."""
import numpy as np
import pytest
import torch

from util_golden import load, sampler_randomness, state_dict

pytestmark = pytest.mark.gpu


class _F:                      # stand-in for a reference field object: only dim() is read by Generator_x0
    def __init__(self, w):
        self.w = int(w)

    def dim(self):
        return self.w


def _nets(z, dev):
    from ddgani_amd.model import Diffusion as D_, Discriminator_model as DM
    widths = z["widths"].tolist()
    D, C, T = int(sum(widths)), len(widths), int(z["T"])
    G = D_.Generator_x0(D, 2 * D, D, T + 1, [_F(w) for w in widths])
    G.load_state_dict(state_dict(z, "G__"))
    Dn = DM.Discriminator_noise_x(2 * D, D // 2, C, T + 1)
    Dn.load_state_dict(state_dict(z, "D__"))
    return G.to(dev).train(), Dn.to(dev).train(), D_.Diffusion_setting(T, D, dev), T


@pytest.mark.parametrize("name", ["sampler_num", "sampler_mixed", "sampler_T10"])
def test_G_D_onestep_against_reference(name):
    """reference: model/Diffusion.py:412-424 with the reference's z and dropout masks -> its x_{T-1}."""
    from ddgani_amd.model import Diffusion as D_
    dev = torch.device("cuda:0")
    z = load(name)
    G, _, DS, T = _nets(z, dev)
    _, masks_b, _ = sampler_randomness(z, T)
    drops = [torch.tensor(m.astype(np.uint8)).to(dev) for m in masks_b[0]]
    x, M = torch.tensor(z["x"]).to(dev), torch.tensor(z["M"]).to(dev)
    tvec = torch.ones(x.shape[0], dtype=torch.long, device=dev) * T          # as the reference passes it (:432)
    out = D_.G_D_onestep(G, DS, torch.tensor(z["x_T"]).to(dev), tvec, M, x, z=torch.tensor(z["z0"]).to(dev), drops=drops)
    np.testing.assert_allclose(out.cpu().numpy(), z["x_Tm1"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("name", ["sampler_num", "sampler_mixed", "sampler_T10"])
def test_reverse_diffusion_G_D_sample_against_reference(name):
    """reference: model/Diffusion.py:426-450 -> (x0_hat, discriminator_out), both against the reference's."""
    from ddgani_amd.model import Diffusion as D_
    dev = torch.device("cuda:0")
    z = load(name)
    G, Dn, DS, T = _nets(z, dev)
    zs, masks_b, _ = sampler_randomness(z, T)
    rnd = D_.ExplicitRandomness(eps_T=torch.tensor(z["eps_T"]).to(dev), zs=[t.to(dev) for t in zs],
                                drops=[[torch.tensor(m.astype(np.uint8)).to(dev) for m in ms] for ms in masks_b])
    x, M = torch.tensor(z["x"]).to(dev), torch.tensor(z["M"]).to(dev)
    x0_hat, d_out = D_.reverse_diffusion_G_D_sample(G, Dn, DS, T, x.shape[0], x.shape[1], dev, x, M, randomness=rnd,
                                                    eps2=torch.tensor(z["eps2"]).to(dev))
    np.testing.assert_allclose(x0_hat.cpu().numpy(), z["x0_hat"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(d_out.cpu().numpy(), z["d_out"], rtol=0, atol=2e-5)


def _corr(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    a, b = a - a.mean(), b - b.mean()
    return float((a * b).sum() / (a.norm() * b.norm()))


def test_default_calls_draw_fresh_noise():
    """A script written against the reference signatures passes no seeds: every call must see new numbers, the z of a
    reverse step must be independent of the eps of forward_diffusion / get_xt, and torch.manual_seed reproduces."""
    from ddgani_amd.model import Diffusion as D_
    dev = torch.device("cuda:0")
    z = load("sampler_num")
    G, Dn, DS, T = _nets(z, dev)
    N = 40000
    D = int(sum(z["widths"].tolist()))
    x = torch.rand(N, D, device=dev)
    M = (torch.rand(N, D, device=dev) < 0.7).float()
    torch.manual_seed(123)
    _, e1 = D_.forward_diffusion(DS, x, T)
    _, e2 = D_.forward_diffusion(DS, x, T)
    xt = D_.get_xt(DS, x, T)
    assert not torch.equal(e1, e2) and abs(_corr(e1, e2)) < 0.01
    s = float(DS.sqrt_one_minus_alpha_cumulative[T - 1])
    e3 = (xt - float(DS.sqrt_alpha_cumulative[T - 1]) * x) / s
    assert abs(_corr(e1, e3)) < 0.01 and abs(_corr(e2, e3)) < 0.01
    # z of G_D_onestep on the observed cells: x_{t-1} = sqrt(abar_{t-1}) x + sqrt(1-abar_{t-1}) z
    a1 = D_.G_D_onestep(G, DS, xt, T, M, x)
    a2 = D_.G_D_onestep(G, DS, xt, T, M, x)
    zz = [(a - float(DS.sqrt_alpha_cumulative[T - 2]) * x) / float(DS.sqrt_one_minus_alpha_cumulative[T - 2]) for a in (a1, a2)]
    obs = M.bool()
    assert abs(zz[0][obs].double().var().item() - 1) < 0.02
    assert abs(_corr(zz[0][obs], zz[1][obs])) < 0.01                      # two calls, two draws
    for e in (e1, e2, e3):
        assert abs(_corr(zz[0][obs], e[obs])) < 0.01                      # z independent of every eps
    # whole sampler: two default passes differ, a re-seeded process state does not matter for freshness
    p1, _ = D_.reverse_diffusion_G_D_sample(G, Dn, DS, T, N, D, dev, x, M)
    p2, _ = D_.reverse_diffusion_G_D_sample(G, Dn, DS, T, N, D, dev, x, M)
    assert not torch.equal(p1, p2)
    # reproducible: same seed and same named call -> same numbers
    r1 = D_.reverse_sample(G.device_params(), T, x, M, G.layout, seed=5, call=77)
    r2 = D_.reverse_sample(G.device_params(), T, x, M, G.layout, seed=5, call=77)
    assert torch.equal(r1, r2)


@pytest.mark.parametrize("widths", [[1, 1, 5, 1, 3, 1, 1, 8, 1], [1] * 177, [4] * 36 + [1] * 5])
def test_fused_head_reverse_step_is_bit_identical(widths):
    """ddg_head_reverse_step == ddg_head followed by ddg_reverse_step, bit for bit, with explicit noise and with the
    on-chip generator (the Philox index is the global element, whatever the tiling); D = 177 gives tiles whose first
    element is not a multiple of 4 (the scalar path)."""
    from ddgani_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(len(widths))
    lay = ops.FieldLayout(widths, dev)
    N, D, T = 5003, lay.D, 6
    y = (torch.randn(N, D, generator=g) * 2).to(dev)
    x_t, x = torch.randn(N, D, generator=g).to(dev), torch.rand(N, D, generator=g).to(dev)
    M = (torch.rand(N, D, generator=g) < 0.7).float().to(dev)
    z = torch.randn(N, D, generator=g).to(dev)
    for t in (T, 3):
        pred = ops.head(y, lay)
        for kw in (dict(z=z), dict(seed=5, stream_id=77)):
            want = ops.reverse_step(x_t, x, M, pred, t, T, **kw)
            got = ops.head_reverse_step(y, x_t, x, M, t, T, lay, **kw)
            assert torch.equal(want, got), (t, list(kw))
    # in place
    xt2 = x_t.clone()
    ops.head_reverse_step(y, xt2, x, M, T, T, lay, z=z, out=xt2)
    assert torch.equal(xt2, ops.reverse_step(x_t, x, M, ops.head(y, lay), T, T, z=z))


@pytest.mark.parametrize("widths", [[1, 1, 5, 1, 3, 1, 1, 8, 1], [1] * 177, [4] * 36 + [1] * 5, [7, 1, 2], [3] * 59])
@pytest.mark.parametrize("rows", [0, 1, 63, 5003])
def test_head_and_mask_merge_against_torch(widths, rows):
    """ddg_head / ddg_mask_merge (model/Diffusion.py:189-201, :598-615) against the same torch ops in fp32 on the device:
    sigmoid on numeric columns, softmax over every categorical block, the observed cells of x kept; tile starts that are
    and are not 16-byte aligned (D = 177, 9), tables with no numeric / no categorical column, empty and one-row inputs."""
    from ddgani_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(len(widths) + rows)
    lay = ops.FieldLayout(widths, dev)
    D = lay.D
    y = (torch.randn(rows, D, generator=g) * 3).to(dev)
    x = torch.rand(rows, D, generator=g).to(dev)
    M = (torch.rand(rows, D, generator=g) < 0.6).float().to(dev)

    def head_ref(v):
        out = torch.empty_like(v)
        for c, w in enumerate(widths):
            a, b = lay.offsets[c], lay.offsets[c + 1]
            out[:, a:b] = torch.softmax(v[:, a:b], dim=1) if w > 1 else torch.sigmoid(v[:, a:b])
        return out

    got = ops.head(y, lay)
    assert got.shape == (rows, D)
    if rows:
        assert float((got - head_ref(y)).abs().max()) <= 3e-7
    merged = ops.mask_merge(x, M, y, lay)
    if rows:
        want = M * x + (1 - M) * y
        for c, w in enumerate(widths):
            if w > 1:
                a, b = lay.offsets[c], lay.offsets[c + 1]
                want[:, a:b] = torch.softmax(want[:, a:b], dim=1)
        assert float((merged - want).abs().max()) <= 3e-7
