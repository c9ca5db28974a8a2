"""Row-sharded (one process per GPU) versions of the hot path.

Each rank owns a contiguous shard of the table's rows.
 * attention fill: queries = the local shard; the masked key rows and their masks are all-gathered
   once (RCCL all-gather over xGMI) so every rank scans all keys; outputs are disjoint.
 * sampler: rows are independent except for train-mode BatchNorm, whose per-layer column sums
   (2*width fp64) are all-reduced before ddg_bn_finalize: 3 tiny all-reduces per generator call.
``backend`` is the op module (ddgani_amd.ops on the GPU; the CPU tests inject the oracle-backed
stand-in from tests/ to exercise the collective logic with gloo).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class Comm:
    def __init__(self, group=None):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.world = dist.get_world_size(group) if self.on else 1
        # gloo (CPU tests, one-GPU rehearsals of the multi-rank path) moves device tensors through the host;
        # nccl (= RCCL, the production backend) works on device memory directly
        self.host_staged = self.on and dist.get_backend(group) == "gloo"

    def all_gather_rows(self, t: torch.Tensor, counts) -> torch.Tensor:
        """concatenate the ranks' row shards (shard sizes ``counts``) along dim 0"""
        if not self.on:
            return t
        if self.host_staged and t.is_cuda:
            dev = t.device
            self.host_staged = False
            try:
                return self.all_gather_rows(t.cpu(), counts).to(dev)
            finally:
                self.host_staged = True
        if len(set(counts)) == 1:
            out = torch.empty((sum(counts),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
            return out
        # uneven shards: collectives need equal sizes, so pad every shard to the largest one
        cmax = max(counts)
        pad = torch.zeros((cmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        out = torch.empty((cmax * len(counts),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, pad, group=self.group)
        return torch.cat([out[r * cmax:r * cmax + c] for r, c in enumerate(counts)], 0)

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.on:
            if self.host_staged and t.is_cuda:
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


# On-chip random numbers are keyed by (seed, stream, LOCAL element index), so what a row draws depends on where its launch
# starts.  To make throughput-mode results independent of the number of ranks, the table is cut into VSHARDS fixed
# VIRTUAL shards: every launch that draws covers one virtual shard and carries that shard's number in its stream id, and a
# rank owns VSHARDS / world consecutive virtual shards -- for world sizes that divide VSHARDS (1, 2, 4, 8: one node) the
# rows, streams and draws are the same whatever the world size.  Other world sizes fall back to one shard per rank.
VSHARDS = 8


def _even_bounds(N: int, parts: int):
    per = (N + parts - 1) // parts
    return [(min(r * per, N), min((r + 1) * per, N)) for r in range(parts)]


def shard_bounds(N: int, world: int):
    """rows of every rank: unions of consecutive virtual shards when ``world`` divides VSHARDS"""
    if VSHARDS % world == 0:
        vb, g = _even_bounds(N, VSHARDS), VSHARDS // world
        return [(vb[r * g][0], vb[(r + 1) * g - 1][1]) for r in range(world)]
    return _even_bounds(N, world)


def virtual_pieces(N: int, world: int, rank: int):
    """[(virtual shard id, lo, hi)] of ``rank``: row ranges RELATIVE to the rank's shard"""
    if VSHARDS % world != 0:
        lo, hi = _even_bounds(N, world)[rank]
        return [(rank, 0, hi - lo)]
    vb, g = _even_bounds(N, VSHARDS), VSHARDS // world
    base = vb[rank * g][0]
    return [(v, vb[v][0] - base, vb[v][1] - base) for v in range(rank * g, (rank + 1) * g) if vb[v][1] > vb[v][0]]


def work_balanced_bounds(m8: torch.Tensor, world: int):
    """Row ranges of the ATTENTION fill's query shards with (nearly) equal work.  The fill computes one similarity per
    (missing cell, key observed in that cell's column), so a row weighs sum over its missing columns of the number of
    keys observed there; even row counts leave most ranks idle when the missing cells sit in a block of rows (Region
    missingness, BASELINE configs[4]: every pair belongs to the ranks that own the rectangle's rows).  ``m8``: the WHOLE
    [N, C] mask (one byte per cell: cheap to all-gather first).  Contiguous ranges that cover all rows; a range may be
    empty.  The sampler keeps shard_bounds (its cost is per row and its draws are tied to the virtual shards)."""
    N = m8.shape[0]
    m = (m8 != 0)
    nobs = m.sum(0, dtype=torch.int64)                                    # keys observed per column
    w = ((~m).to(torch.int64) * nobs[None, :]).sum(1)                     # work per query row
    c = torch.cumsum(w, 0)
    total = int(c[-1].item()) if N else 0
    if total == 0:
        return _even_bounds(N, world)
    targets = torch.tensor([total * r // world for r in range(1, world)], dtype=torch.int64, device=c.device)
    cuts = [0] + (torch.searchsorted(c, targets, right=False) + 1).clamp_(max=N).tolist() + [N]
    for i in range(1, len(cuts)):                                         # (monotone by construction; keep it explicit)
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def sharded_attn_fill(comm: Comm, backend, X0_local, m_local, W, layout, k: int, N_total: int, out=None,
                      query_bounds=None):
    """top-k attention fill against ALL rows.  Every rank contributes its shard_bounds rows (``X0_local``, ``m_local``) to
    the all-gather of the keys.  Its queries are those same rows by default; with ``query_bounds`` (work_balanced_bounds:
    tables whose missing cells are not spread over the rows) they are rows query_bounds[rank] of the gathered table, and
    the result covers THOSE rows -- the key exchange stays an even all-gather whatever the query ranges are."""
    bounds = shard_bounds(N_total, comm.world)
    counts = [b - a for a, b in bounds]
    if not comm.on:
        return backend.attn_topk_fill(X0_local, m_local, X0_local, m_local, W, layout, k, out=out)
    Xk = comm.all_gather_rows(X0_local, counts)
    mk = comm.all_gather_rows(m_local, counts)
    lo, hi = bounds[comm.rank] if query_bounds is None else query_bounds[comm.rank]
    if query_bounds is None:
        Xq, mq = X0_local, m_local
    else:
        Xq, mq = Xk[lo:hi], mk[lo:hi]
    if hi == lo:                                                      # no query on this rank
        return out if out is not None else Xq.clone()
    return backend.attn_topk_fill(Xq, mq, Xk, mk, W, layout, k, q_row_offset=lo, out=out)


def _cut(a, lo, hi):
    return None if a is None else a[lo:hi]


def _into(dst, res):
    """``res`` is what a backend call returned for ``out=dst``: the HIP backend wrote into dst and hands it back (nothing
    to do), the CPU stand-in of the tests returns a fresh tensor (copy it in)"""
    if res.data_ptr() != dst.data_ptr():
        dst.copy_(res)


def sharded_denoiser_fwd(comm: Comm, backend, P, x_t, t, layout, N_total: int, drops=None, drop_p: float = 0.2,
                         seed: int = 0, stream_id: int = 0, row_offset: int = 0, bn_eps: float = 1e-5, out=None,
                         pieces=None, apply_head: bool = True):
    """Generator_x0.forward over a row shard with global (sync) BatchNorm statistics.  ``pieces``: [(lo, hi, stream_id)]
    row ranges of the shard with the stream number each one draws its dropout from (the virtual shards of
    sharded_reverse_sample); default: the whole shard under ``stream_id``."""
    w = P.t
    dev = x_t.device
    d = [None, None, None] if drops is None else drops
    rows = x_t.shape[0]
    if pieces is None:
        pieces = [(0, rows, stream_id)]
    h = x_t
    pro = backend.PRO_ADD_EMB
    scale = shift = None
    for l, (li, bi) in enumerate(((0, 1), (4, 5), (8, 9))):
        Wl, bl = w[f"linears.{li}.weight"], w[f"linears.{li}.bias"]
        stats = torch.zeros(2 * Wl.shape[0], dtype=torch.float64, device=dev)
        z = torch.empty((rows, Wl.shape[0]), dtype=torch.float32, device=dev)
        for lo, hi, sid in pieces:
            if l == 0:
                kw = dict(emb=w["time_embedding.weight"], t=t if not isinstance(t, torch.Tensor) else t[lo:hi])
            else:
                kw = dict(scale=scale, shift=shift, drop=_cut(d[l - 1], lo, hi), drop_p=drop_p, seed=seed,
                          stream_id=sid * 4 + l - 1)
            _into(z[lo:hi], backend.linear(h[lo:hi], Wl, bl, prologue=pro, stats=stats, out=z[lo:hi], **kw))
        comm.all_reduce_sum(stats)
        scale, shift = backend.bn_finalize(stats, w[f"linears.{bi}.weight"], w[f"linears.{bi}.bias"], float(N_total),
                                           bn_eps)
        h = z
        pro = backend.PRO_BN_RELU
    y = out if out is not None else torch.empty((rows, w["linears.12.weight"].shape[0]), dtype=torch.float32, device=dev)
    for lo, hi, sid in pieces:
        _into(y[lo:hi], backend.linear(h[lo:hi], w["linears.12.weight"], w["linears.12.bias"], prologue=pro, out=y[lo:hi],
                                       scale=scale, shift=shift, drop=_cut(d[2], lo, hi), drop_p=drop_p, seed=seed,
                                       stream_id=sid * 4 + 2))
    return backend.head(y, layout, out=y) if apply_head else y


def sharded_reverse_sample(comm: Comm, backend, P, T: int, x, M, layout, N_total: int, randomness=None, seed: int = 0,
                           drop_p: float = 0.2, call: Optional[int] = None):
    """T-step reverse diffusion over a row shard (model/Diffusion.py:426-450), sync-BN across ranks.
    On-chip draws: one launch per VIRTUAL shard, whose number (not the rank) goes into the stream id
    (model.Diffusion.stream_of), so the result does not depend on the world size (1, 2, 4, 8 ranks); every rank must
    pass the same ``call`` (default: its own next_call(), equal on all ranks when they make the same calls in the same
    order)."""
    from .model.Diffusion import KIND_DROP, KIND_EPS, KIND_Z, drop_seed, next_call, stream_of
    r = randomness
    call = next_call() if call is None else call
    vp = virtual_pieces(N_total, comm.world, comm.rank)
    x_t = torch.empty_like(x)
    for v, lo, hi in vp:
        _into(x_t[lo:hi], backend.forward_noise(x[lo:hi], T, T, eps=None if r is None else r.eps_T[lo:hi], seed=seed,
                                                stream_id=stream_of(call, KIND_EPS, T, v), out=x_t[lo:hi]))
    pred = None
    i = 0
    for ts in range(T, 1, -1):
        pred = sharded_denoiser_fwd(comm, backend, P, x_t, ts, layout, N_total,
                                    drops=None if r is None else r.drops[i], drop_p=drop_p, seed=drop_seed(seed),
                                    pieces=[(lo, hi, stream_of(call, KIND_DROP, ts, v)) for v, lo, hi in vp])
        if ts > 2:
            for v, lo, hi in vp:
                _into(x_t[lo:hi], backend.reverse_step(x_t[lo:hi], x[lo:hi], M[lo:hi], pred[lo:hi], ts, T,
                                                       z=None if r is None else r.zs[i][lo:hi], seed=seed,
                                                       stream_id=stream_of(call, KIND_Z, ts, v), out=x_t[lo:hi]))
        i += 1
    return pred


def batch_bounds(batch: int, world: int):
    """rows of the global minibatch owned by each rank (contiguous, sizes differ by at most one, never empty for
    batch >= world)"""
    base, rem = divmod(batch, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def sharded_train_step(comm: Comm, backend, which: str, PG, PD, state, x0, M, m, t, layout, sched_dev, T: int,
                       loss_weight=None, loss_scale: float = 1.0, eps=None, drops=None, seed: int = 0,
                       stream_id: int = 0):
    """One data-parallel minibatch step (model/Diffusion.py:287-311 generator, :356-372 discriminator): every rank
    holds the same GLOBAL batch (x0, M, m, t and, in parity mode, eps / drops: the same indices and draws on every
    rank), computes the gradients of its row share with synchronised BatchNorm (ddg_train_*_grads_dp all-reduces the
    BatchNorm sums through ``comm``), all-reduces the flat gradient buffer once (RCCL: ``nccl`` backend) and takes the
    identical Adam step.  ``state``: model.Diffusion._NetState of the network being trained.
    -> loss values of the global batch (device tensor: [res, g_d] or [d_loss])."""
    B = x0.shape[0]
    lo, hi = batch_bounds(B, comm.world)[comm.rank]
    cut = lambda a: None if a is None else a[lo:hi].contiguous()
    dp = backend.DataParallel(B, comm.all_reduce_sum) if comm.on else None
    kw = dict(eps=cut(eps), drops=None if drops is None else [cut(d) for d in drops], seed=seed, stream_id=stream_id,
              dp=dp)
    if which == "G":
        losses, _ = backend.train_gen_grads(PG, PD, state.buf, cut(x0), cut(M), cut(m), cut(t), layout, sched_dev, T,
                                            loss_weight["Res_weight"], loss_weight["G_D_weight"], **kw)
    else:
        losses, _ = backend.train_disc_grads(PG, PD, state.buf, cut(x0), cut(M), cut(m), cut(t), layout, sched_dev, T,
                                             loss_scale=loss_scale, **kw)
    if comm.on:
        comm.all_reduce_sum(state.buf.flat)          # gradients + loss values in one collective
    state.step()
    return losses.clone()
