"""GPU: the training steps replayed as HIP graphs (model/Diffusion.py::GraphedSteps) against the eager entry points: same
batch rows, timesteps and random-stream numbers -> bit-identical gradients, on the first (eager + capture) call and on a
replay, whose stream number exists only in device memory (ddg_set_stream_offset)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_graphed_steps_equal_eager():
    from ddgani_amd import ops, synth
    from ddgani_amd.model import Diffusion as D_, Discriminator_model as DM
    dev = torch.device("cuda:0")
    tab = synth.make_table("c3", dev, rows=20_000)
    lay, T = tab.layout, tab.T

    class F:
        def __init__(s, w): s.w = w; s.data_type = "Categorical Data" if w > 1 else "Numerical Data"
        def dim(s): return s.w
    torch.manual_seed(3407)
    G = D_.Generator_x0(lay.D, 2 * lay.D, lay.D, T + 1, [F(w) for w in lay.widths]).to(dev)
    Dn = DM.Discriminator_noise_x(2 * lay.D, lay.D // 2, lay.C, T + 1).to(dev)
    PG, PD = G.device_params(refresh=True), Dn.device_params(refresh=True)
    bufG, bufD = ops.ParamGrads(PG, ops.GEN_KEYS), ops.ParamGrads(PD, ops.DISC_KEYS)
    DS = D_.Diffusion_setting(T, lay.D, dev)
    m_data = tab.m8.float()
    seed, base = 5, D_.GraphedSteps.BASE_STREAM

    def g_fn(x0, Mb, mb, t, stream=base):
        return ops.train_gen_grads(PG, PD, bufG, x0, Mb, mb, t, lay, DS.table, T, 100.0, 1.0, seed=seed, stream_id=stream)[0]

    def d_fn(x0, Mb, mb, t, stream=base):
        return ops.train_disc_grads(PG, PD, bufD, x0, Mb, mb, t, lay, DS.table, T, loss_scale=1.0, seed=seed,
                                    stream_id=stream)[0]

    # eager references with the draws the graphed path will make (same generator seed, same order)
    gen = np.random.default_rng(11)
    ref = []
    for i in range(4):                       # steps: G, D, G (replay), D (replay)
        idx = torch.as_tensor(gen.choice(tab.x_diff.shape[0], 128, replace=False, shuffle=False), device=dev)
        t = torch.as_tensor(gen.integers(2, T + 1, size=128), device=dev)
        fn, buf = (g_fn, bufG) if i % 2 == 0 else (d_fn, bufD)
        # the library derives several streams from the step's number (x 8 + 7, x 4 + layer ...) and adds the device word to
        # each of them, so the reference sets the same word instead of folding it into the argument
        ctr = torch.tensor([i << 32], dtype=torch.int64, device=dev)
        ops.set_stream_offset(ctr)
        try:
            loss = fn(tab.x_diff[idx].contiguous(), tab.M[idx].contiguous(), m_data[idx].contiguous(), t)
            torch.cuda.synchronize()
        finally:
            ops.set_stream_offset(None)
        ref.append((loss.clone(), buf.flat.clone()))
    rnd = D_.ChipRandom(11)
    gs = D_.GraphedSteps(tab.x_diff, tab.M, m_data, T, dev)
    try:
        for i in range(4):
            fn, buf = (g_fn, bufG) if i % 2 == 0 else (d_fn, bufD)
            loss = gs.step("G" if i % 2 == 0 else "D", fn, rnd)
            torch.cuda.synchronize()
            assert torch.equal(loss, ref[i][0]), (i, loss, ref[i][0])
            assert torch.equal(buf.flat, ref[i][1]), i
        assert int(gs.counter.item()) == 4 << 32
    finally:
        gs.close()


def test_stream_offset_is_an_addition_to_the_stream_number():
    """on-chip noise with stream s and device offset c == stream s + c without offset; None switches it off"""
    from ddgani_amd import ops
    dev = torch.device("cuda:0")
    x0 = torch.rand(1000, 24, device=dev)
    plain = ops.forward_noise(x0, 4, 6, seed=3, stream_id=1077)
    ctr = torch.tensor([77], dtype=torch.int64, device=dev)
    ops.set_stream_offset(ctr)
    try:
        shifted = ops.forward_noise(x0, 4, 6, seed=3, stream_id=1000)
        torch.cuda.synchronize()
    finally:
        ops.set_stream_offset(None)
    again = ops.forward_noise(x0, 4, 6, seed=3, stream_id=1000)
    assert torch.equal(plain, shifted) and not torch.equal(plain, again)
