"""GPU: the training steps replayed as HIP graphs (model/Diffusion.py::GraphedSteps) against the eager entry points: same
batch rows, timesteps and random-stream numbers -> bit-identical gradients, on the first (eager + capture) call and on a
replay, whose stream number exists only in device memory (ddg_set_stream_offset); several hundred back-to-back steps
WITHOUT host synchronisation (how the training loop runs them) bit-identical to the eager loop; no memset / memcpy node
in the captured steps (the cause of round 2's NaN defect)."""
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


def _hospital_like(dev, rows=800):
    """a table shaped like the reference's hospital run: D = 145 (odd: the layer kernels' element-wise operand path),
    seven one-hot fields and two numeric ones, MCAR 0.2"""
    from ddgani_amd import ops
    widths = [19, 2, 40, 30, 5, 17, 30, 1, 1]
    lay = ops.FieldLayout(widths, dev)
    g = torch.Generator(device="cpu"); g.manual_seed(7)
    x = torch.zeros(rows, lay.D)
    for c, w in enumerate(widths):
        a = lay.offsets[c]
        if w == 1:
            x[:, a] = torch.rand(rows, generator=g)
        else:
            x[torch.arange(rows), a + torch.randint(0, w, (rows,), generator=g)] = 1.0
    m8 = (torch.rand(rows, len(widths), generator=g) < 0.8)
    M = torch.repeat_interleave(m8.float(), torch.tensor(widths), dim=1)
    return lay, x.to(dev).contiguous(), M.to(dev).contiguous(), m8.float().to(dev).contiguous()


def _nets(lay, T, dev):
    from ddgani_amd import ops
    from ddgani_amd.model import Diffusion as D_, Discriminator_model as DM

    class F:
        def __init__(s, w): s.w = w; s.data_type = "Categorical Data" if w > 1 else "Numerical Data"
        def dim(s): return s.w
    torch.manual_seed(3407)
    G = D_.Generator_x0(lay.D, 2 * lay.D, lay.D, T + 1, [F(w) for w in lay.widths]).to(dev)
    Dn = DM.Discriminator_noise_x(2 * lay.D, lay.D // 2, lay.C, T + 1).to(dev)
    optG = D_._NetState(G.device_params(refresh=True), ops.GEN_KEYS, 1e-4, 1e-4)
    optD = D_._NetState(Dn.device_params(refresh=True), ops.DISC_KEYS, 1e-4, 0.0)
    return optG, optD


@pytest.mark.parametrize("graph_g", [True, False])
def test_unsynchronised_replays_equal_eager_loop(graph_g):
    """40 epochs of 6 generator + 6 discriminator steps with Adam after each, no host synchronisation between steps
    (one device-to-host read per epoch, as the training loop has): the graphed run ends with bit-identical weights,
    gradients and loss history to the eager run that uses the same rows, timesteps and stream numbers.
    graph_g=False is the configuration in which round 2's defect showed (--UseFD: eager generator steps between graphed
    discriminator steps)."""
    from ddgani_amd import ops
    from ddgani_amd.model import Diffusion as D_
    dev = torch.device("cuda:0")
    lay, x, M, m = _hospital_like(dev)
    T, epochs, seed, base = 6, 40, 3047, D_.GraphedSteps.BASE_STREAM
    DS = D_.Diffusion_setting(T, lay.D, dev)

    def fns(optG, optD):
        def g_fn(x0, Mb, mb, t):
            return ops.train_gen_grads(optG.P, optD.P, optG.buf, x0, Mb, mb, t, lay, DS.table, T, 100.0, 1.0, seed=seed,
                                       stream_id=base)[0]

        def d_fn(x0, Mb, mb, t):
            return ops.train_disc_grads(optG.P, optD.P, optD.buf, x0, Mb, mb, t, lay, DS.table, T, loss_scale=1.0,
                                        seed=seed, stream_id=base)[0]
        return g_fn, d_fn

    def eager_step(fn, gen, ctr):
        idx = torch.as_tensor(gen.choice(x.shape[0], 128, replace=False, shuffle=False), device=dev)
        t = torch.as_tensor(gen.integers(2, T + 1, size=128), device=dev)
        loss = fn(x.index_select(0, idx), M.index_select(0, idx), m.index_select(0, idx), t)
        ctr.add_(1 << 32)
        return loss

    # ---- eager reference
    optG, optD = _nets(lay, T, dev)
    g_fn, d_fn = fns(optG, optD)
    gen = np.random.default_rng(11)
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    ops.set_stream_offset(ctr)
    ref_hist = []
    try:
        for _ in range(epochs):
            ls = []
            for fn, opt in [(g_fn, optG)] * T + [(d_fn, optD)] * T:
                ls.append(eager_step(fn, gen, ctr).clone().flatten()[:1])
                opt.step()
            ref_hist.append(torch.cat(ls).cpu())
        torch.cuda.synchronize()
    finally:
        ops.set_stream_offset(None)
    ref = ({k: v.clone() for k, v in optG.P.t.items()}, {k: v.clone() for k, v in optD.P.t.items()},
           optG.buf.flat.clone(), optD.buf.flat.clone())
    assert all(bool(torch.isfinite(h).all()) for h in ref_hist)

    # ---- graphed run (generator steps eager when graph_g is False; they read the same offset word)
    optG, optD = _nets(lay, T, dev)
    g_fn, d_fn = fns(optG, optD)
    rnd = D_.ChipRandom(11)
    gs = D_.GraphedSteps(x, M, m, T, dev)
    hist = []
    try:
        for _ in range(epochs):
            ls = []
            for name, fn, opt in [("G", g_fn, optG)] * T + [("D", d_fn, optD)] * T:
                if name == "G" and not graph_g:
                    torch.add(gs.counter, 0, out=gs.offset)
                    l = eager_step(fn, rnd.gen, gs.counter)
                    gs.offset.mul_(0)
                else:
                    l = gs.step(name, fn, rnd)
                opt.step()
                ls.append(l.clone().flatten()[:1])
            hist.append(torch.cat(ls).cpu())
        torch.cuda.synchronize()
        assert int(gs.counter.item()) == (epochs * 2 * T) << 32 and int(gs.offset.item()) == 0
    finally:
        gs.close()
    for e, (a, b) in enumerate(zip(hist, ref_hist)):
        assert torch.equal(a, b), (e, a, b)
    for k in ref[0]:
        assert torch.equal(optG.P.t[k], ref[0][k]), k
    for k in ref[1]:
        assert torch.equal(optD.P.t[k], ref[1][k]), k
    assert torch.equal(optG.buf.flat, ref[2]) and torch.equal(optD.buf.flat, ref[3])


def test_plain_calls_are_not_shifted_while_a_trainer_is_open():
    """the offset word is zero outside an executing step: (seed, stream_id) of any other call keeps its meaning; a
    second live instance is refused"""
    from ddgani_amd import ops
    from ddgani_amd.model import Diffusion as D_
    dev = torch.device("cuda:0")
    lay, x, M, m = _hospital_like(dev, rows=300)
    x0 = torch.rand(500, 24, device=dev)
    plain = ops.forward_noise(x0, 4, 6, seed=3, stream_id=1077)
    optG, optD = _nets(lay, 6, dev)
    DS = D_.Diffusion_setting(6, lay.D, dev)
    gs = D_.GraphedSteps(x, M, m, 6, dev)
    try:
        with pytest.raises(RuntimeError):
            D_.GraphedSteps(x, M, m, 6, dev)
        rnd = D_.ChipRandom(1)
        for _ in range(3):
            gs.step("D", lambda a, b, c, t: ops.train_disc_grads(optG.P, optD.P, optD.buf, a, b, c, t, lay, DS.table, 6,
                                                                 seed=1, stream_id=D_.GraphedSteps.BASE_STREAM)[0], rnd)
        assert torch.equal(ops.forward_noise(x0, 4, 6, seed=3, stream_id=1077), plain)
    finally:
        gs.close()
    gs2 = D_.GraphedSteps(x, M, m, 6, dev)         # after close() a new instance is fine
    gs2.close()


def test_captured_steps_hold_no_memset_or_memcpy_node(tmp_path):
    """Round 2's NaN: a hipMemsetAsync NODE of the replayed discriminator step (the clear of the denoiser's BatchNorm
    sums) filled parts of the sums with garbage once in about five runs.  The library zeroes and copies with kernels
    now; this test walks the node types of the captured graph (hipGraphGetNodes) and fails if a memset / memcpy node
    comes back -- in the library's entry points or in GraphedSteps' own bookkeeping."""
    from ddgani_amd import ops
    from ddgani_amd.model import Diffusion as D_
    dev = torch.device("cuda:0")
    lay, x, M, m = _hospital_like(dev, rows=300)
    optG, optD = _nets(lay, 6, dev)
    DS = D_.Diffusion_setting(6, lay.D, dev)
    idx = torch.arange(128, device=dev)
    t = torch.full((128,), 3, dtype=torch.int64, device=dev)
    x0, Mb, mb = x[idx].contiguous(), M[idx].contiguous(), m[idx].contiguous()

    def body():
        ops.train_gen_grads(optG.P, optD.P, optG.buf, x0, Mb, mb, t, lay, DS.table, 6, 100.0, 1.0, seed=1, stream_id=5)
        ops.train_disc_grads(optG.P, optD.P, optD.buf, x0, Mb, mb, t, lay, DS.table, 6, seed=1, stream_id=6)
        ops.denoiser_fwd(optG.P, x0, 3, lay, seed=1, stream_id=7)
    body()
    torch.cuda.synchronize()
    types = _captured_node_types(body)
    assert types.count(0) >= 40, types                      # hipGraphNodeTypeKernel
    assert set(types) == {0}, f"non-kernel nodes in the captured steps (1 = memcpy, 2 = memset): {sorted(set(types))}"
    gs = D_.GraphedSteps(x, M, m, 6, dev)
    try:
        d_fn = lambda a, b, c, tt: ops.train_disc_grads(optG.P, optD.P, optD.buf, a, b, c, tt, lay, DS.table, 6, seed=1,
                                                        stream_id=D_.GraphedSteps.BASE_STREAM)[0]
        gs._body(d_fn)
        torch.cuda.synchronize()
        types = _captured_node_types(lambda: gs._body(d_fn))
        assert set(types) == {0}, f"non-kernel nodes in GraphedSteps._body: {sorted(set(types))}"
    finally:
        gs.close()


def _captured_node_types(body):
    """hipGraphNodeType of every node of the graph that torch captures from body()"""
    import ctypes
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        body()
    hip = ctypes.CDLL("libamdhip64.so")
    raw = ctypes.c_void_p(g.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    assert hip.hipGraphGetNodes(raw, None, ctypes.byref(n)) == 0
    nodes = (ctypes.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) == 0
    out = []
    for i in range(n.value):
        ty = ctypes.c_int(-1)
        assert hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(ty)) == 0
        out.append(ty.value)
    return out


def test_node_type_walk_sees_a_memset_node():
    """the check above can see what it guards against: torch's zero_() on a large tensor is captured as a memset node"""
    dev = torch.device("cuda:0")
    buf = torch.ones(1 << 20, device=dev)
    other = torch.ones(1 << 20, device=dev)

    def body():
        buf.zero_()
        other.copy_(buf)
    body()
    torch.cuda.synchronize()
    types = _captured_node_types(body)
    assert any(t in (1, 2) for t in types), types
