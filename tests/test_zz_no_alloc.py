"""GPU: ddg_denoiser_fwd_static (the entry point that allocates nothing: static device scratch) against
ddg_denoiser_fwd (caller workspace) through the C-ABI: two calls in a row, the second on a smaller input,
then the size limit.  Collected last (file name) so that a failure here cannot hide another test."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(rows, lay, P, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    D = lay.D
    x_t = torch.randn(rows, D, generator=g).to(dev)
    t = torch.randint(2, 7, (rows,), generator=g).to(dev)
    drops = [(torch.rand(rows, w, generator=g) < 0.8).to(torch.uint8).to(dev) for w in (2 * D, D, 2 * D)]
    return x_t, t, drops


def test_static_scratch_entry_point_matches_workspace_entry_point():
    from ddgani_amd import ops, synth, _lib
    dev = torch.device("cuda:0")
    widths = [1, 1, 5, 1, 3, 1, 1, 8, 1, 1, 1, 2]
    lay = ops.FieldLayout(widths, dev)
    P = ops.GenParams(synth.random_generator_state(lay.D, 6), dev)
    # the array keeps the last call's contents: poison check = a large call first, then a smaller one
    # (9 000 rows: the branch of 8 192 rows and more -- split-f16 layer kernels, weights pre-split INTO the static array)
    for rows, seed in ((9000, 5), (3000, 1), (777, 2), (128, 3)):
        x_t, t, drops = _case(rows, lay, P, dev, seed)
        for tt in (t, 4):
            want = ops.denoiser_fwd(P, x_t, tt, lay, drops=drops)
            got = ops.denoiser_fwd_static(P, x_t, tt, lay, drops=drops)
            torch.cuda.synchronize()
            assert torch.isfinite(got).all()
            # same kernels on the same inputs; only the order of the fp64 BatchNorm-sum atomics may differ
            assert (got - want).abs().max().item() <= 1e-6, (rows, (got - want).abs().max().item())
    # on-chip dropout (Philox) path: identical streams -> same result
    x_t, t, _ = _case(500, lay, P, dev, 4)
    want = ops.denoiser_fwd(P, x_t, 3, lay, seed=11, stream_id=5)
    got = ops.denoiser_fwd_static(P, x_t, 3, lay, seed=11, stream_id=5)
    assert (got - want).abs().max().item() <= 1e-6


def test_static_scratch_rejects_oversized_input():
    from ddgani_amd import ops, synth, _lib
    dev = torch.device("cuda:0")
    lay = ops.FieldLayout([1] * 64, dev)
    P = ops.GenParams(synth.random_generator_state(lay.D, 6), dev)
    L = _lib.lib()
    limit = 32 << 20
    rows = 1
    while L.ddg_denoiser_workspace_bytes(rows, 64, 128, 64) <= limit:
        rows *= 2
    x_t = torch.zeros(rows, 64, device=dev)
    with pytest.raises(_lib.DdgError, match="static scratch"):
        ops.denoiser_fwd_static(P, x_t, 3, lay, drop_p=0.0)
    ok = rows // 2
    out = ops.denoiser_fwd_static(P, x_t[:ok].contiguous(), 3, lay, drop_p=0.0)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
