"""CPU, world_size 2 over gloo: the row-sharded hot path (ddgani_amd/dist.py) gives every rank the
rows it would get from the single-process path.  The collective logic (key-row all-gather, BatchNorm
statistic all-reduce) is what is under test: the arithmetic is served by an oracle-backed stand-in for
``ddgani_amd.ops`` (test infrastructure), because the HIP library needs a GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleOps:
    """Stand-in with the call surface dist.py uses from ddgani_amd.ops, computed by the CPU oracle."""
    PRO_NONE, PRO_ADD_EMB, PRO_BN_RELU, PRO_EMB_RELU = 0, 1, 2, 3

    def __init__(self):
        from oracle import ddgani_oracle as orc
        self.orc = orc

    def attn_topk_fill(self, Xq0, mq, Xk0, mk, W, layout, k, q_row_offset=0, out=None):
        # whole-table closed form on the gathered keys, then the caller's query rows
        X0 = Xk0.double().numpy()
        data_m = mk.numpy().astype(np.int64)
        cf, _ = self.orc.attn_fill_closed_form(torch.tensor(X0), W, data_m, layout.offsets, k)
        return torch.tensor(cf[q_row_offset:q_row_offset + Xq0.shape[0]], dtype=torch.float32)

    def linear(self, A, W, bias, prologue=0, emb=None, t=None, scale=None, shift=None, drop=None, drop_p=0.0, seed=0,
               stream_id=0, stats=None, out=None):
        h = A
        if prologue == self.PRO_ADD_EMB:
            h = A + emb[t]
        elif prologue == self.PRO_BN_RELU:
            h = torch.relu(A * scale + shift)
            if drop is not None:
                h = h * drop.float() / (1 - drop_p)
        y = torch.nn.functional.linear(h, W, bias)
        if stats is not None:
            stats[:y.shape[1]] += y.double().sum(0)
            stats[y.shape[1]:] += (y.double() ** 2).sum(0)
        return y

    def bn_finalize(self, stats, gamma, beta, total_rows, eps=1e-5):
        w = gamma.shape[0]
        mean = stats[:w] / total_rows
        var = (stats[w:] / total_rows - mean * mean).clamp_min(0)
        sc = gamma.double() / torch.sqrt(var + eps)
        return sc.float(), (beta.double() - mean * sc).float()

    def head(self, y, layout, out=None):
        return self.orc.head(y, layout.offsets, layout.is_cat)

    def forward_noise(self, x0, t, T, eps=None, seed=0, stream_id=0, out=None):
        S = self.orc.schedule(T)
        return self.orc.forward_noise(S, x0, torch.full((x0.shape[0],), t, dtype=torch.long), eps)

    def reverse_step(self, x_t, x, M, pred, t, T, z=None, seed=0, stream_id=0, out=None):
        S = self.orc.schedule(T)
        return self.orc.reverse_step(S, x_t, x, M, pred, z, torch.full((x.shape[0],), t, dtype=torch.long))


class _Layout:
    def __init__(self, widths):
        self.widths = list(widths)
        self.offsets = [0]
        for w in widths:
            self.offsets.append(self.offsets[-1] + w)
        self.is_cat = [w > 1 for w in widths]
        self.C, self.D = len(widths), self.offsets[-1]


class _P:
    def __init__(self, sd):
        self.t = sd


def _case():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util_golden import load, sampler_randomness, state_dict
    z = load("sampler_mixed")
    T = int(z["T"])
    zs, masks_b, _ = sampler_randomness(z, T)
    return z, T, zs, masks_b, state_dict(z, "G__")


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception as e:          # surface the failure instead of leaving the parent waiting
        import traceback
        q.put((rank, "error", traceback.format_exc()))
        raise


def _worker_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ddgani_amd import dist as ddist
    z, T, zs, masks_b, sd = _case()
    lay = _Layout(z["widths"].tolist())
    N = z["x"].shape[0]
    lo, hi = ddist.shard_bounds(N, world)[rank]
    comm = ddist.Comm()
    be = OracleOps()
    x, M = torch.tensor(z["x"]), torch.tensor(z["M"])

    class R:
        eps_T = torch.tensor(z["eps_T"])[lo:hi]
        zs_ = [t[lo:hi] for t in zs]
        drops = [[torch.tensor(m.astype(np.uint8))[lo:hi] for m in ms] for ms in masks_b]
    R.zs = R.zs_
    with torch.no_grad():
        out = ddist.sharded_reverse_sample(comm, be, _P(sd), T, x[lo:hi], M[lo:hi], lay, N, randomness=R)
        # attention fill on the same table (uneven shards: 300 rows over 2 ranks is even, so use 257)
        za = np.load(os.path.join(ROOT, "tests", "golden", "attn_mixed257.npz"))
        la = _Layout(np.diff(za["offsets"]).tolist())
        Na = za["X"].shape[0]
        alo, ahi = ddist.shard_bounds(Na, world)[rank]
        Mx = np.repeat(za["data_m"], np.diff(za["offsets"]), axis=1).astype(np.float32)
        X0 = torch.tensor(za["X"] * Mx)
        m8 = torch.tensor(za["data_m"].astype(np.uint8))
        code = ddist.sharded_attn_fill(comm, be, X0[alo:ahi], m8[alo:ahi], torch.tensor(za["W"]), la, int(za["k"]), Na)
        # the same fill with the queries sharded by work: the keys still come from the even shards, the result covers
        # the rank's row range of work_balanced_bounds
        qb = ddist.work_balanced_bounds(m8, world)
        code_w = ddist.sharded_attn_fill(comm, be, X0[alo:ahi], m8[alo:ahi], torch.tensor(za["W"]), la, int(za["k"]), Na,
                                         query_bounds=qb)
    q.put((rank, lo, hi, out.numpy(), alo, ahi, code.numpy(), qb[rank][0], qb[rank][1], code_w.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_sampler_and_attention_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for r in res:
        assert r[1] != "error", r[2]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    z, T, _, _, _ = _case()
    za = np.load(os.path.join(ROOT, "tests", "golden", "attn_mixed257.npz"))
    covered = 0
    covered_w = 0
    for rank, lo, hi, out, alo, ahi, code, qlo, qhi, code_w in sorted(res, key=lambda r: r[0]):
        # sync-BN over the two shards == BatchNorm over the whole table (golden x0_hat of the reference)
        np.testing.assert_allclose(out, z["x0_hat"][lo:hi], rtol=0, atol=2e-5)
        np.testing.assert_allclose(code, za["new_code"][alo:ahi], rtol=0, atol=1e-5)
        assert code_w.shape[0] == qhi - qlo
        np.testing.assert_allclose(code_w, za["new_code"][qlo:qhi], rtol=0, atol=1e-5)
        covered += hi - lo
        covered_w += qhi - qlo
    assert covered == z["x"].shape[0] and covered_w == za["X"].shape[0]


def test_shard_bounds_cover_rows():
    from ddgani_amd.dist import shard_bounds
    for N, w in ((10, 3), (257, 2), (8, 8), (5, 8), (1000000, 8)):
        b = shard_bounds(N, w)
        assert b[0][0] == 0 and b[-1][1] == N and all(b[i][1] == b[i + 1][0] for i in range(w - 1))


def test_virtual_shards_partition_rows_identically_for_every_world_size():
    """ranks own 8 / world consecutive virtual shards: the (virtual shard, global row range) pairs are the same set for
    world sizes 1, 2, 4, 8, so on-chip draws (keyed by virtual shard and row inside it) do not depend on the world size"""
    sys.path.insert(0, ROOT)
    from ddgani_amd import dist as ddist
    for N in (1, 7, 8, 30_001, 1_000_000):
        ref = None
        for world in (1, 2, 4, 8):
            bounds = ddist.shard_bounds(N, world)
            assert bounds[0][0] == 0 and bounds[-1][1] == N and all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
            pieces = []
            for r in range(world):
                for v, lo, hi in ddist.virtual_pieces(N, world, r):
                    pieces.append((v, bounds[r][0] + lo, bounds[r][0] + hi))
            ref = ref or pieces
            assert pieces == ref, (N, world)
    # a world size that does not divide 8: one piece per rank, the rank as its number
    assert ddist.virtual_pieces(100, 3, 1) == [(1, 0, 34)]


def test_work_balanced_bounds_even_out_a_region_rectangle():
    """Attention queries sharded by work (missing cells x observed keys of their column): contiguous ranges that cover all
    rows; with the missing cells in one block of rows (Region missingness) every rank gets about the same work, where even
    row counts would give it all to one or two ranks; with cells missing everywhere the ranges are close to even."""
    from ddgani_amd import dist as ddist
    g = torch.Generator().manual_seed(3)
    N, C = 20000, 16

    def work(m8, lo, hi):
        m = m8 != 0
        nobs = m.sum(0, dtype=torch.int64)
        return int(((~m[lo:hi]).to(torch.int64) * nobs[None, :]).sum())
    region = torch.ones(N, C, dtype=torch.uint8)
    region[3000:5500, 4:12] = 0
    mcar = (torch.rand(N, C, generator=g) < 0.7).to(torch.uint8)
    for m8, name in ((region, "region"), (mcar, "mcar")):
        for world in (1, 2, 4, 8):
            b = ddist.work_balanced_bounds(m8, world)
            assert len(b) == world and b[0][0] == 0 and b[-1][1] == N
            assert all(b[r][1] == b[r + 1][0] for r in range(world - 1)) and all(lo <= hi for lo, hi in b)
            ws = [work(m8, lo, hi) for lo, hi in b]
            total = sum(ws)
            assert total == work(m8, 0, N)
            # every share within one row's largest possible work (C * N) per cut of the ideal one
            assert max(ws) <= total / world + C * N and min(ws) >= total / world - 2 * C * N, (name, world, ws)
        if name == "region":
            even = [work(m8, lo, hi) for lo, hi in ddist.shard_bounds(N, 8)]
            assert max(even) > 3 * total / 8                                   # what even row counts would do
        else:
            b = ddist.work_balanced_bounds(m8, 8)
            assert all(abs((hi - lo) - N / 8) < 0.02 * N for lo, hi in b)
    # nothing missing: even rows
    assert ddist.work_balanced_bounds(torch.ones(64, 4, dtype=torch.uint8), 4) == [(0, 16), (16, 32), (32, 48), (48, 64)]
