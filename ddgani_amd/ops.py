"""torch-tensor front end of the C-ABI (include/ddgani.h).

PyTorch is used for device memory, streams and nothing else: every function hands raw device
pointers of contiguous tensors to libddgani_hip.so on torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import DiscGrads, DiscWeights, GenGrads, GenWeights, TrainBatch, check, lib

PRO_NONE, PRO_ADD_EMB, PRO_BN_RELU, PRO_EMB_RELU = 0, 1, 2, 3
ATTN_F32, ATTN_BF16_FILTER = 0, 1
MAX_TOPK_REG = 30      # DDG_MAX_TOPK_REG (include/ddgani.h): larger k takes the radix-select path
_NAMES8 = ("betas", "alphas", "sqrt_alphas", "one_by_sqrt_alpha_s", "sqrt_beta_s", "alpha_cumulative",
           "sqrt_alpha_cumulative", "sqrt_one_minus_alpha_cumulative")


def _ptr(t: Optional[torch.Tensor], dtype=None) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.DdgError("ddgani_amd ops need tensors on the GPU (no CPU fallback)")
    if not t.is_contiguous():
        raise _lib.DdgError("tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise _lib.DdgError(f"expected {dtype}, got {t.dtype}")
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def set_stream_offset(counter: Optional[torch.Tensor]):
    """ddg_set_stream_offset: every on-chip random draw on the current device uses stream_id + counter[0] (an int64 / uint64
    device tensor the caller keeps alive and advances between graph replays); None switches it off."""
    if counter is not None and (counter.dtype not in (torch.int64, torch.uint64) or counter.numel() < 1):
        raise _lib.DdgError("stream offset: one 64-bit integer on the device")
    check(lib().ddg_set_stream_offset(_ptr(counter)), "ddg_set_stream_offset")


def gather_batch(x, M, m, idx, out=None, counter=None, offset=None):
    """ddg_gather_batch: (x[idx], M[idx], m[idx]) in one launch; ``out``: the three preallocated destinations;
    counter / offset: the captured step's stream-offset words (see model.Diffusion.GraphedSteps)."""
    B, D, Cn = idx.numel(), x.shape[1], m.shape[1]
    if out is None:
        out = (torch.empty((B, D), dtype=torch.float32, device=x.device), torch.empty((B, D), dtype=torch.float32, device=x.device),
               torch.empty((B, Cn), dtype=torch.float32, device=x.device))
    check(lib().ddg_gather_batch(_ptr(x, torch.float32), _ptr(M, torch.float32), _ptr(m, torch.float32), _ptr(idx, torch.int64),
                                 x.shape[0], B, D, Cn, _ptr(out[0], torch.float32), _ptr(out[1], torch.float32),
                                 _ptr(out[2], torch.float32), _ptr(counter), _ptr(offset), _stream()), "ddg_gather_batch")
    return out


def step_end(counter, offset):
    check(lib().ddg_step_end(_ptr(counter), _ptr(offset), _stream()), "ddg_step_end")


class FieldLayout:
    """Encoded-column layout of a table: column c occupies [offsets[c], offsets[c+1])."""

    def __init__(self, widths: Sequence[int], device=None):
        self.widths = [int(w) for w in widths]
        self.offsets = [0]
        for w in self.widths:
            self.offsets.append(self.offsets[-1] + w)
        self.is_cat = [w > 1 for w in self.widths]
        self.C = len(self.widths)
        self.D = self.offsets[-1]
        self._dev = {}
        self.offsets_host = (C.c_int32 * (self.C + 1))(*self.offsets)
        if device is not None:
            self.on(device)

    def on(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.tensor(self.offsets, dtype=torch.int32, device=device),
                              torch.tensor(self.is_cat, dtype=torch.uint8, device=device))
        return self._dev[key]


# --------------------------------------------------------------------------- a3
def schedule(T: int) -> np.ndarray:
    """[8,T] fp32: rows in the order of ddg_schedule (Diffusion_setting, model/Diffusion.py:93-108)."""
    out = np.zeros((8, T), dtype=np.float32)
    check(lib().ddg_schedule(T, out.ctypes.data_as(C.POINTER(C.c_float))), "ddg_schedule")
    return out


def schedule_dict(T: int):
    s = schedule(T)
    return {n: s[i] for i, n in enumerate(_NAMES8)}


# --------------------------------------------------------------------------- a5
def forward_noise(x0: torch.Tensor, t, T: int, eps: Optional[torch.Tensor] = None, seed: int = 0,
                  stream_id: int = 0, sched_dev: Optional[torch.Tensor] = None, return_eps: bool = False, out=None):
    rows, D = x0.shape
    x_t = torch.empty_like(x0) if out is None else out
    eps_out = torch.empty_like(x0) if (return_eps and eps is None) else None
    if isinstance(t, torch.Tensor):
        if sched_dev is None:
            sched_dev = torch.from_numpy(schedule(T)).to(x0.device)
        tp, tu = _ptr(t, torch.int64), 0
    else:
        tp, tu = None, int(t)
    check(lib().ddg_forward_noise(_ptr(x0, torch.float32), _ptr(eps, torch.float32), tp, tu,
                                  _ptr(sched_dev, torch.float32), T, seed, stream_id, _ptr(x_t), _ptr(eps_out),
                                  rows, D, _stream()), "ddg_forward_noise")
    if return_eps:
        return x_t, (eps if eps is not None else eps_out)
    return x_t


# --------------------------------------------------------------------------- a7
def reverse_step(x_t, x, M, pred, t: int, T: int, z=None, seed: int = 0, stream_id: int = 0, out=None,
                 sched_host: Optional[np.ndarray] = None):
    rows, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    sh = schedule(T) if sched_host is None else sched_host
    check(lib().ddg_reverse_step(_ptr(x_t, torch.float32), _ptr(x, torch.float32), _ptr(M, torch.float32),
                                 _ptr(pred, torch.float32), _ptr(z, torch.float32),
                                 sh.ctypes.data_as(C.POINTER(C.c_float)), T, int(t), seed, stream_id, _ptr(out),
                                 rows, D, _stream()), "ddg_reverse_step")
    return out


def head_reverse_step(y, x_t, x, M, t: int, T: int, layout: "FieldLayout", z=None, seed: int = 0, stream_id: int = 0,
                      out=None, sched_host: Optional[np.ndarray] = None):
    """ddg_head_reverse_step: reverse step from the generator's pre-head logits ``y`` (head fused in)."""
    rows, D = x.shape
    off, cat = layout.on(x.device)
    if out is None:
        out = torch.empty_like(x)
    sh = schedule(T) if sched_host is None else sched_host
    check(lib().ddg_head_reverse_step(_ptr(y, torch.float32), _ptr(x_t, torch.float32), _ptr(x, torch.float32),
                                      _ptr(M, torch.float32), _ptr(z, torch.float32),
                                      sh.ctypes.data_as(C.POINTER(C.c_float)), T, int(t), seed, stream_id, _ptr(off),
                                      _ptr(cat), layout.C, _ptr(out), rows, D, _stream()), "ddg_head_reverse_step")
    return out


# --------------------------------------------------------------------------- a6 head / a9
def head(y, layout: FieldLayout, out=None):
    rows, D = y.shape
    off, cat = layout.on(y.device)
    if out is None:
        out = torch.empty_like(y)
    check(lib().ddg_head(_ptr(y, torch.float32), _ptr(off), _ptr(cat), layout.C, _ptr(out), rows, D, _stream()),
          "ddg_head")
    return out


def mask_merge(x, M, xhat, layout: FieldLayout, out=None):
    rows, D = x.shape
    off, cat = layout.on(x.device)
    if out is None:
        out = torch.empty_like(x)
    check(lib().ddg_mask_merge(_ptr(x, torch.float32), _ptr(M, torch.float32), _ptr(xhat, torch.float32), _ptr(off),
                               _ptr(cat), layout.C, _ptr(out), rows, D, _stream()), "ddg_mask_merge")
    return out


# --------------------------------------------------------------------------- dense layer
def linear(A, W, bias, prologue=PRO_NONE, A2=None, emb=None, t=None, scale=None, shift=None, drop=None,
           drop_p: float = 0.0, seed: int = 0, stream_id: int = 0, stats=None, out=None):
    rows = A.shape[0]
    in1 = A.shape[1]
    inn = in1 + (A2.shape[1] if A2 is not None else 0)
    outd = W.shape[0]
    assert W.shape[1] == inn
    if out is None:
        out = torch.empty((rows, outd), dtype=torch.float32, device=A.device)
    if isinstance(t, torch.Tensor):
        tp, tu = _ptr(t, torch.int64), 0
    else:
        tp, tu = None, int(t or 0)
    check(lib().ddg_linear(_ptr(A, torch.float32), _ptr(A2, torch.float32), in1, _ptr(W, torch.float32),
                           _ptr(bias, torch.float32), prologue, _ptr(emb, torch.float32), tp, tu,
                           _ptr(scale, torch.float32), _ptr(shift, torch.float32), _ptr(drop, torch.uint8),
                           float(drop_p), seed, stream_id, _ptr(out), _ptr(stats, torch.float64), rows, inn, outd,
                           _stream()), "ddg_linear")
    return out


def bn_finalize(stats, gamma, beta, total_rows: float, eps: float = 1e-5):
    width = gamma.shape[0]
    scale = torch.empty(width, dtype=torch.float32, device=gamma.device)
    shift = torch.empty_like(scale)
    check(lib().ddg_bn_finalize(_ptr(stats, torch.float64), _ptr(gamma, torch.float32), _ptr(beta, torch.float32),
                                float(total_rows), eps, _ptr(scale), _ptr(shift), width, _stream()),
          "ddg_bn_finalize")
    return scale, shift


# --------------------------------------------------------------------------- a6 / a10 whole nets
class GenParams:
    """Device copy of a Generator_x0 state_dict (key layout of model/Diffusion.py:148-178)."""

    def __init__(self, sd, device):
        g = lambda k: sd[k].detach().to(device=device, dtype=torch.float32).contiguous()
        self.t = {k: g(k) for k in ("linears.0.weight", "linears.0.bias", "linears.1.weight", "linears.1.bias",
                                    "linears.4.weight", "linears.4.bias", "linears.5.weight", "linears.5.bias",
                                    "linears.8.weight", "linears.8.bias", "linears.9.weight", "linears.9.bias",
                                    "linears.12.weight", "linears.12.bias", "time_embedding.weight")}
        t = self.t
        self.in_dim = t["linears.0.weight"].shape[1]
        self.enc_dim = t["linears.0.weight"].shape[0]
        self.lat_dim = t["linears.4.weight"].shape[0]
        self.n_steps = t["time_embedding.weight"].shape[0]
        self.c = GenWeights(self.in_dim, self.enc_dim, self.lat_dim, self.n_steps,
                            *[t[k].data_ptr() for k in ("linears.0.weight", "linears.0.bias", "linears.1.weight",
                                                        "linears.1.bias", "linears.4.weight", "linears.4.bias",
                                                        "linears.5.weight", "linears.5.bias", "linears.8.weight",
                                                        "linears.8.bias", "linears.9.weight", "linears.9.bias",
                                                        "linears.12.weight", "linears.12.bias",
                                                        "time_embedding.weight")])


class DiscParams:
    """Device copy of a Discriminator_noise_x state_dict (model/Discriminator_model.py:85-102)."""

    def __init__(self, sd, device):
        g = lambda k: sd[k].detach().to(device=device, dtype=torch.float32).contiguous()
        keys = ("linears.0.weight", "linears.0.bias", "linears.2.weight", "linears.2.bias", "linears.4.weight",
                "linears.4.bias", "linears.6.weight", "linears.6.bias", "step_embeddings.0.weight",
                "step_embeddings.1.weight", "step_embeddings.2.weight")
        self.t = {k: g(k) for k in keys}
        t = self.t
        self.in_dim = t["linears.0.weight"].shape[1]
        self.lat_dim = t["linears.0.weight"].shape[0]
        self.out_dim = t["linears.6.weight"].shape[0]
        self.n_steps = t["step_embeddings.0.weight"].shape[0]
        self.c = DiscWeights(self.in_dim, self.lat_dim, self.out_dim, self.n_steps, *[t[k].data_ptr() for k in keys])


_ws_cache = {}


def _workspace(nbytes: int, device, tag: str) -> torch.Tensor:
    key = (str(device), tag)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        if os.environ.get("DDG_DEBUG_POISON"):       # every float / double of a fresh workspace reads as NaN
            ws.fill_(0xFF)
        _ws_cache[key] = ws
    return ws


def denoiser_fwd(P: GenParams, x_t, t, layout: FieldLayout, drops=None, drop_p: float = 0.2, seed: int = 0,
                 stream_id: int = 0, bn_eps: float = 1e-5, apply_head: bool = True, out=None):
    rows = x_t.shape[0]
    nb = lib().ddg_denoiser_workspace_bytes(rows, P.in_dim, P.enc_dim, P.lat_dim)
    ws = _workspace(nb, x_t.device, "den")
    off, cat = layout.on(x_t.device)
    if out is None:
        out = torch.empty_like(x_t)
    if isinstance(t, torch.Tensor):
        tp, tu = _ptr(t, torch.int64), 0
    else:
        tp, tu = None, int(t)
    d = [None, None, None] if drops is None else [_ptr(m, torch.uint8) for m in drops]
    check(lib().ddg_denoiser_fwd(C.byref(P.c), _ptr(x_t, torch.float32), tp, tu, d[0], d[1], d[2], float(drop_p),
                                 seed, stream_id, bn_eps, _ptr(off), _ptr(cat), layout.C, int(apply_head), _ptr(out),
                                 rows, ws.data_ptr(), ws.numel(), _stream()), "ddg_denoiser_fwd")
    return out


def denoiser_fwd_static(P: GenParams, x_t, t, layout: FieldLayout, drops=None, drop_p: float = 0.2, seed: int = 0,
                        stream_id: int = 0, bn_eps: float = 1e-5, apply_head: bool = True, out=None):
    """ddg_denoiser_fwd_static: the denoiser forward without a workspace (static device scratch, no allocation
    inside the library; ``out`` should be passed by callers that must not allocate at all)."""
    rows = x_t.shape[0]
    off, cat = layout.on(x_t.device)
    if out is None:
        out = torch.empty_like(x_t)
    if isinstance(t, torch.Tensor):
        tp, tu = _ptr(t, torch.int64), 0
    else:
        tp, tu = None, int(t)
    d = [None, None, None] if drops is None else [_ptr(m, torch.uint8) for m in drops]
    check(lib().ddg_denoiser_fwd_static(C.byref(P.c), _ptr(x_t, torch.float32), tp, tu, d[0], d[1], d[2],
                                        float(drop_p), seed, stream_id, bn_eps, _ptr(off), _ptr(cat), layout.C,
                                        int(apply_head), _ptr(out), rows, _stream()), "ddg_denoiser_fwd_static")
    return out


def disc_fwd(P: DiscParams, xtj1, xt, t, out=None):
    rows = xt.shape[0]
    nb = lib().ddg_disc_workspace_bytes(rows, P.lat_dim)
    ws = _workspace(nb, xt.device, "disc")
    if out is None:
        out = torch.empty((rows, P.out_dim), dtype=torch.float32, device=xt.device)
    if isinstance(t, torch.Tensor):
        tp, tu = _ptr(t, torch.int64), 0
    else:
        tp, tu = None, int(t)
    check(lib().ddg_disc_fwd(C.byref(P.c), _ptr(xtj1, torch.float32), _ptr(xt, torch.float32), tp, tu, _ptr(out), rows,
                             ws.data_ptr(), ws.numel(), _stream()), "ddg_disc_fwd")
    return out


# --------------------------------------------------------------------------- a11-a15 training step
GEN_KEYS = ("linears.0.weight", "linears.0.bias", "linears.1.weight", "linears.1.bias",
            "linears.4.weight", "linears.4.bias", "linears.5.weight", "linears.5.bias",
            "linears.8.weight", "linears.8.bias", "linears.9.weight", "linears.9.bias",
            "linears.12.weight", "linears.12.bias", "time_embedding.weight")
DISC_KEYS = ("linears.0.weight", "linears.0.bias", "linears.2.weight", "linears.2.bias", "linears.4.weight",
             "linears.4.bias", "linears.6.weight", "linears.6.bias", "step_embeddings.0.weight",
             "step_embeddings.1.weight", "step_embeddings.2.weight")


def _batch_struct(x0, M, m, t, eps, drops, drop_p, seed, stream_id) -> TrainBatch:
    d = [None, None, None] if drops is None else [_ptr(q, torch.uint8) for q in drops]
    return TrainBatch(_ptr(x0, torch.float32), _ptr(M, torch.float32), _ptr(m, torch.float32), _ptr(t, torch.int64),
                      _ptr(eps, torch.float32), d[0], d[1], d[2], float(drop_p), int(seed), int(stream_id),
                      x0.shape[0])


class ParamGrads:
    """Gradient (and Adam moment) buffers shaped like a GenParams / DiscParams tensor dict.  The gradients are views
    of ONE flat buffer followed by ``n_extra`` floats (the step's loss values), so a data-parallel step completes them
    with a single all-reduce (``flat``)."""

    def __init__(self, P, keys, n_extra: int = 2):
        self.keys = keys
        self.P = P
        sizes = [P.t[k].numel() for k in keys]
        dev = P.t[keys[0]].device
        self.flat = torch.zeros(sum(sizes) + n_extra, dtype=torch.float32, device=dev)
        self.g, o = {}, 0
        for k, n in zip(keys, sizes):
            self.g[k] = self.flat[o:o + n].view_as(P.t[k])
            o += n
        self.extra = self.flat[o:]
        self.exp_avg = {k: torch.zeros_like(P.t[k]) for k in keys}
        self.exp_avg_sq = {k: torch.zeros_like(P.t[k]) for k in keys}
        self.step = 0
        n = len(keys)
        arr = lambda d: (C.c_void_p * n)(*[d[k].data_ptr() for k in keys])
        self._p, self._g, self._m, self._v = arr(P.t), arr(self.g), arr(self.exp_avg), arr(self.exp_avg_sq)
        self._n = (C.c_int64 * n)(*[P.t[k].numel() for k in keys])

    def adam_step(self, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        """torch.optim.Adam.step() on the device tensors of P (model/Diffusion.py:571-573)."""
        self.step += 1
        check(lib().ddg_adam_step(len(self.keys), self._p, self._g, self._m, self._v, self._n, self.step, lr,
                                  betas[0], betas[1], eps, weight_decay, _stream()), "ddg_adam_step")


class DataParallel:
    """struct ddg_dp for one training-step call: ``allreduce(tensor)`` sums a device tensor in place over the ranks
    (ddgani_amd.dist.Comm.all_reduce_sum).  The library names the buffers by device pointer; they lie inside the
    step's workspace tensor, so the callback hands torch a typed view of that tensor."""

    def __init__(self, global_rows: int, allreduce):
        self.global_rows = int(global_rows)
        self.allreduce = allreduce
        self.ws = None
        self.error = None

        def cb(ctx, buf, count, is_f64, stream):
            try:
                off = int(buf) - self.ws.data_ptr()
                nb = int(count) * (8 if is_f64 else 4)
                if off < 0 or off + nb > self.ws.numel():
                    raise _lib.DdgError("all-reduce buffer outside the step workspace")
                self.allreduce(self.ws[off:off + nb].view(torch.float64 if is_f64 else torch.float32))
                return 0
            except Exception as e:          # never unwind through the C frame
                self.error = e
                return 1
        self._cb = _lib.ALLREDUCE_FN(cb)
        self.c = _lib.Dp(self.global_rows, self._cb, None)

    def bind(self, ws):
        self.ws = ws
        self.error = None
        return C.byref(self.c)


def _train_ws(PG, PD, rows, C_, device):
    nb = lib().ddg_train_workspace_bytes(rows, PG.in_dim, PG.enc_dim, PG.lat_dim, PD.lat_dim, C_)
    return _workspace(nb, device, "train")


def train_gen_grads(PG: "GenParams", PD: "DiscParams", grads: ParamGrads, x0, M, m, t, layout: FieldLayout,
                    sched_dev, T: int, w_res: float, w_gd: float, eps=None, drops=None, drop_p: float = 0.2,
                    seed: int = 0, stream_id: int = 0, bn_eps: float = 1e-5, want_pred: bool = False,
                    dp: Optional[DataParallel] = None):
    """Generator step up to the gradients (train_one_epoch_Generator, model/Diffusion.py:287-310).
    Returns (losses[2] device tensor = res, g_d; pred or None).  With ``dp`` the rows are this rank's share of the
    global batch and gradients / losses are this rank's share of the global ones (ddg_train_gen_grads_dp);
    the losses land in ``grads.extra`` so that one all-reduce of ``grads.flat`` completes everything."""
    dev = x0.device
    off, cat = layout.on(dev)
    ws = _train_ws(PG, PD, x0.shape[0], layout.C, dev)
    losses = grads.extra[:2] if dp is not None else torch.empty(2, dtype=torch.float32, device=dev)
    pred = torch.empty_like(x0) if want_pred else None
    b = _batch_struct(x0, M, m, t, eps, drops, drop_p, seed, stream_id)
    gs = GenGrads(*[grads.g[k].data_ptr() for k in GEN_KEYS])
    rc = lib().ddg_train_gen_grads_dp(C.byref(PG.c), C.byref(PD.c), C.byref(b), _ptr(sched_dev, torch.float32), T,
                                      _ptr(off), _ptr(cat), layout.C, float(w_res), float(w_gd), bn_eps, C.byref(gs),
                                      _ptr(losses), _ptr(pred), ws.data_ptr(), ws.numel(), _stream(),
                                      dp.bind(ws) if dp is not None else None)
    if dp is not None and dp.error is not None:
        raise dp.error
    check(rc, "ddg_train_gen_grads")
    return losses, pred


def train_disc_grads(PG: "GenParams", PD: "DiscParams", grads: ParamGrads, x0, M, m, t, layout: FieldLayout,
                     sched_dev, T: int, loss_scale: float = 1.0, eps=None, drops=None, drop_p: float = 0.2,
                     seed: int = 0, stream_id: int = 0, bn_eps: float = 1e-5, want_pred: bool = False,
                     dp: Optional[DataParallel] = None):
    """Discriminator step up to the gradients (train_one_epoch_discriminator_noise_x, :356-370)."""
    dev = x0.device
    off, cat = layout.on(dev)
    ws = _train_ws(PG, PD, x0.shape[0], layout.C, dev)
    loss = grads.extra[:1] if dp is not None else torch.empty(1, dtype=torch.float32, device=dev)
    pred = torch.empty_like(x0) if want_pred else None
    b = _batch_struct(x0, M, m, t, eps, drops, drop_p, seed, stream_id)
    gs = DiscGrads(*[grads.g[k].data_ptr() for k in DISC_KEYS])
    rc = lib().ddg_train_disc_grads_dp(C.byref(PG.c), C.byref(PD.c), C.byref(b), _ptr(sched_dev, torch.float32), T,
                                       _ptr(off), _ptr(cat), layout.C, float(loss_scale), bn_eps, C.byref(gs),
                                       _ptr(loss), _ptr(pred), ws.data_ptr(), ws.numel(), _stream(),
                                       dp.bind(ws) if dp is not None else None)
    if dp is not None and dp.error is not None:
        raise dp.error
    check(rc, "ddg_train_disc_grads")
    return loss, pred


# --------------------------------------------------------------------------- a1 + a2
def attn_topk_fill(Xq0, mq, Xk0, mk, W, layout: FieldLayout, k: int, q_row_offset: int = 0, debug: bool = False,
                   out=None, mode: int = ATTN_BF16_FILTER):
    """Row-attention top-k fill (ddg_attn_topk_fill).  Xq0/Xk0: masked encoded rows [N,D] fp32;
    mq/mk: uint8 [N,C]; W: [C,D].  Returns new_code [Nq,D] (+ tau, nvalid when debug)."""
    Nq, D = Xq0.shape
    Nk = Xk0.shape[0]
    Cc = layout.C
    nb = lib().ddg_attn_workspace_bytes(Nq, Nk, Cc, D, int(k))
    if nb == 0:
        raise _lib.DdgError(f"attention fill: unsupported shape D={D} (max {256})")
    ws = _workspace(nb, Xq0.device, "attn")
    if out is None:
        out = torch.empty_like(Xq0)
    tau = nval = None
    if debug:
        tau = torch.full((Nq, Cc), float("nan"), dtype=torch.float32, device=Xq0.device)
        nval = torch.full((Nq, Cc), -1, dtype=torch.int32, device=Xq0.device)
    check(lib().ddg_attn_topk_fill(_ptr(Xq0, torch.float32), _ptr(mq, torch.uint8), q_row_offset,
                                   _ptr(Xk0, torch.float32), _ptr(mk, torch.uint8), _ptr(W, torch.float32),
                                   layout.offsets_host, Nq, Nk, Cc, D, int(k), int(mode), _ptr(out), _ptr(tau), _ptr(nval),
                                   ws.data_ptr(), ws.numel(), _stream()), "ddg_attn_topk_fill")
    if debug:
        return out, tau, nval
    return out


class AttnPlan:
    """The three phases of the attention fill on one workspace (ddg_attn_prepare / _select / _fill)."""

    def __init__(self, Xq0, mq, Xk0, mk, W, layout: FieldLayout, k: int, q_row_offset: int = 0,
                 mode: int = ATTN_BF16_FILTER):
        self.mode = int(mode)
        self.Xq0, self.mq, self.Xk0, self.mk, self.W, self.layout, self.k = Xq0, mq, Xk0, mk, W, layout, int(k)
        self.q_row_offset = q_row_offset
        self.Nq, self.D = Xq0.shape
        self.Nk = Xk0.shape[0]
        self.alias = int(Xq0.data_ptr() == Xk0.data_ptr() and mq.data_ptr() == mk.data_ptr() and self.Nq == self.Nk)
        nb = lib().ddg_attn_workspace_bytes(self.Nq, self.Nk, layout.C, self.D, self.k)
        if nb == 0:
            raise _lib.DdgError(f"attention fill: unsupported shape D={self.D}")
        self.ws = _workspace(nb, Xq0.device, "attn")

    def prepare(self):
        check(lib().ddg_attn_prepare(_ptr(self.Xq0, torch.float32), _ptr(self.mq, torch.uint8), self.q_row_offset,
                                     _ptr(self.Xk0, torch.float32), _ptr(self.mk, torch.uint8),
                                     _ptr(self.W, torch.float32), self.layout.offsets_host, self.Nq, self.Nk,
                                     self.layout.C, self.D, self.k, self.mode, self.ws.data_ptr(), self.ws.numel(),
                                     _stream()), "ddg_attn_prepare")

    def select(self, out=None):
        """one similarity pass; k <= 30 also fills every pair without a three-way tie at the k-th logit"""
        if out is None:
            out = torch.empty_like(self.Xq0)
        self.out = out
        check(lib().ddg_attn_select(_ptr(self.Xq0, torch.float32), self.Nq, self.Nk, self.layout.C, self.D, self.k,
                                    self.mode, self.alias, _ptr(out), self.ws.data_ptr(), self.ws.numel(), _stream()),
              "ddg_attn_select")
        return out

    def fill(self, debug: bool = False):
        """the pairs select() left open (tie-flagged, or all of them when k > 30)"""
        tau = nval = None
        if debug:
            tau = torch.full((self.Nq, self.layout.C), float("nan"), dtype=torch.float32, device=self.out.device)
            nval = torch.full((self.Nq, self.layout.C), -1, dtype=torch.int32, device=self.out.device)
        check(lib().ddg_attn_fill(self.layout.offsets_host, self.Nq, self.Nk, self.layout.C, self.D, self.k,
                                  self.mode, self.alias, _ptr(self.out), _ptr(tau), _ptr(nval), self.ws.data_ptr(),
                                  self.ws.numel(), _stream()), "ddg_attn_fill")
        return (self.out, tau, nval) if debug else self.out


def attn_set_key_chunk(positions: int):
    """ddg_attn_set_key_chunk (tests / tuning): fixed key-chunk length of the screening path, 0 = default, -1 = environment"""
    check(lib().ddg_attn_set_key_chunk(int(positions)), "ddg_attn_set_key_chunk")


def attn_fallback_pairs(Nq: int, Nk: int, layout: FieldLayout, k: int, alias: bool, device) -> int:
    """ddg_attn_fallback_pairs: pairs of the last fill on the cached attention workspace that took the exact fp32
    fallback (synchronises)."""
    ws = _ws_cache[(str(device), "attn")]
    out = torch.zeros(1, dtype=torch.int32, device=device)
    check(lib().ddg_attn_fallback_pairs(Nq, Nk, layout.C, layout.D, int(k), int(alias), out.data_ptr(), ws.data_ptr(),
                                        ws.numel(), _stream()), "ddg_attn_fallback_pairs")
    return int(out.item())


# ----------------------------------------------------------------------------- 8(f) row 1: Learner
def learner_train(x_train, y_train, x_valid, y_valid, W, b, epochs: int, lr: float = 0.01):
    """``epochs`` full-batch SGD steps of the reference's Learner (model/Learner.py:43-67) on the device, then its
    validation loss and accuracy.  W [L, D] / b [L]: fc2 parameters, updated in place.
    -> (val_loss: 0-dim device tensor, accuracy: float)"""
    L, D = W.shape
    dev = W.device
    xt = x_train.detach().to(torch.float32).contiguous()
    xv = x_valid.detach().to(torch.float32).contiguous()
    yt = y_train.reshape(-1).to(torch.int32).contiguous()
    yv = y_valid.reshape(-1).to(torch.int32).contiguous()
    if xt.shape[1] != D or xv.shape[1] != D or yt.numel() != xt.shape[0] or yv.numel() != xv.shape[0]:
        raise _lib.DdgError("learner_train: shape mismatch")
    ws = torch.empty(lib().ddg_learner_workspace_bytes(D, L), dtype=torch.uint8, device=dev)
    out_loss = torch.empty(1, dtype=torch.float32, device=dev)
    out_hit = torch.empty(1, dtype=torch.int32, device=dev)
    check(lib().ddg_learner_train(_ptr(xt, torch.float32), _ptr(yt, torch.int32), xt.shape[0], _ptr(xv, torch.float32),
                                  _ptr(yv, torch.int32), xv.shape[0], D, L, int(epochs), float(lr),
                                  _ptr(W, torch.float32), _ptr(b, torch.float32), _ptr(out_loss), _ptr(out_hit),
                                  ws.data_ptr(), ws.numel(), _stream()), "ddg_learner_train")
    return out_loss[0], int(out_hit.item()) / xv.shape[0]


DPLAIN_KEYS = ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")


class DplainParams:
    """Device tensors of the plain discriminator ``D`` (model/Discriminator_model.py:5-39: fc1, fc2, fc3), shaped for
    ParamGrads / ddg_adam_step like GenParams."""

    def __init__(self, sd, device):
        self.t = {k: sd[k].detach().to(device=device, dtype=torch.float32).contiguous() for k in DPLAIN_KEYS}
        self.in_dim, self.out_dim = self.t["fc1.weight"].shape[1], self.t["fc3.weight"].shape[0]
        self.c = _lib.DplainWeights(self.in_dim, self.out_dim, *[self.t[k].data_ptr() for k in DPLAIN_KEYS])


def dplain_grads(P: DplainParams, grads: ParamGrads, x, m, drops=None, drop_p: float = 0.2, seed: int = 0,
                 stream_id: int = 0):
    """ddg_dplain_grads: one full-batch step of get_valid_data_index's discriminator up to the gradients."""
    rows = x.shape[0]
    ws = _workspace(lib().ddg_dplain_workspace_bytes(rows, P.in_dim, P.out_dim), x.device, "dplain")
    g = _lib.DplainWeights(P.in_dim, P.out_dim, *[grads.g[k].data_ptr() for k in DPLAIN_KEYS])
    d = [None, None] if drops is None else [_ptr(q, torch.uint8) for q in drops]
    check(lib().ddg_dplain_grads(C.byref(P.c), _ptr(x, torch.float32), _ptr(m, torch.float32), d[0], d[1], float(drop_p),
                                 int(seed), int(stream_id), rows, C.byref(g), ws.data_ptr(), ws.numel(), _stream()),
          "ddg_dplain_grads")


def dplain_fwd(P: DplainParams, x, drops=None, drop_p: float = 0.2, seed: int = 0, stream_id: int = 0):
    """ddg_dplain_fwd -> p [rows, out_dim] (train-mode forward: dropout live)."""
    rows = x.shape[0]
    ws = _workspace(lib().ddg_dplain_workspace_bytes(rows, P.in_dim, P.out_dim), x.device, "dplain")
    out = torch.empty(rows, P.out_dim, dtype=torch.float32, device=x.device)
    d = [None, None] if drops is None else [_ptr(q, torch.uint8) for q in drops]
    check(lib().ddg_dplain_fwd(C.byref(P.c), _ptr(x, torch.float32), d[0], d[1], float(drop_p), int(seed), int(stream_id),
                               rows, _ptr(out), ws.data_ptr(), ws.numel(), _stream()), "ddg_dplain_fwd")
    return out


# ----------------------------------------------------------------------------- 8(f) row 2: column-association tables
CODE_MISSING = 65535      # DDG_CODE_MISSING


def contingency_tables(codes: np.ndarray, obs: np.ndarray, card: Sequence[int], device):
    """Contingency tables of every column pair i < j over the rows observed in both (ddg_contingency; the C^2
    ``pd.crosstab`` calls of utils/get_attr_attn.py:61-77).  codes [N,C] integer codes (host), obs [N,C] bool,
    card[c] = number of codes of column c.  -> dict {(i, j): int64 table [card_i, card_j]} on the host."""
    N, Cn = codes.shape
    if max(card) >= CODE_MISSING:
        raise _lib.DdgError("contingency_tables: more than 65534 distinct values in a column")
    ct = np.where(obs, codes, CODE_MISSING).astype(np.uint16).T.copy()            # [C][N], column-major
    pi, pj = np.triu_indices(Cn, k=1)
    sizes = np.asarray(card, dtype=np.int64)[pi] * np.asarray(card, dtype=np.int64)[pj]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    dev = torch.device(device)
    up = lambda a, dt: torch.tensor(np.ascontiguousarray(a), dtype=dt).to(dev)
    d_ct = torch.from_numpy(ct.view(np.int16)).to(dev)
    d_card, d_pi, d_pj, d_off = up(card, torch.int32), up(pi, torch.int32), up(pj, torch.int32), up(off, torch.int64)
    tables = torch.empty(max(int(off[-1]), 1), dtype=torch.int32, device=dev)
    check(lib().ddg_contingency(d_ct.data_ptr(), N, Cn, _ptr(d_card), _ptr(d_pi), _ptr(d_pj), _ptr(d_off), len(pi),
                                int(off[-1]), tables.data_ptr(), _stream()), "ddg_contingency")
    flat = tables.cpu().numpy().astype(np.int64)
    return {(int(i), int(j)): flat[off[p]:off[p + 1]].reshape(card[i], card[j]) for p, (i, j) in enumerate(zip(pi, pj))}


# ----------------------------------------------------------------------------- 8(f) row 3: decode + score on the device
def score_table(code, ori_norm, m8, layout: FieldLayout, dec_K, dec_min, dec_f32, pos_value, norm_min, norm_den):
    """ddg_score_table -> sums [C,3] fp64 on the device (sum d^2, sum |d|, mismatches over the missing cells)."""
    N, D = code.shape
    off, cat = layout.on(code.device)
    sums = torch.zeros(layout.C, 3, dtype=torch.float64, device=code.device)
    check(lib().ddg_score_table(_ptr(code, torch.float32), _ptr(ori_norm, torch.float64), _ptr(m8, torch.uint8), _ptr(off),
                                _ptr(cat), _ptr(dec_K, torch.float64), _ptr(dec_min, torch.float64),
                                _ptr(dec_f32, torch.uint8), _ptr(pos_value, torch.float64), _ptr(norm_min, torch.float64),
                                _ptr(norm_den, torch.float64), _ptr(sums), N, layout.C, D, _stream()), "ddg_score_table")
    return sums


def encode_table(vals, layout: FieldLayout, mode, scale, shift, out=None):
    """ddg_encode_table: vals [N,C] fp64 (numeric value / categorical position) -> code [N,D] fp32.
    mode uint8 [C] (0 = min-max, 1 = z-score), scale / shift fp64 [C]; all on the device."""
    N = vals.shape[0]
    off, cat = layout.on(vals.device)
    if out is None:
        out = torch.empty(N, layout.D, dtype=torch.float32, device=vals.device)
    check(lib().ddg_encode_table(_ptr(vals, torch.float64), _ptr(off), _ptr(cat), _ptr(mode, torch.uint8),
                                 _ptr(scale, torch.float64), _ptr(shift, torch.float64), _ptr(out, torch.float32), N,
                                 layout.C, layout.D, _stream()), "ddg_encode_table")
    return out


def decode_table(code, m8, obs, layout: FieldLayout, dec_K, dec_min, dec_f32, pos_value):
    """ddg_decode_table -> [N,C] fp64 on the device: observed cells of ``obs``, decoded cells elsewhere."""
    N = code.shape[0]
    off, cat = layout.on(code.device)
    out = torch.empty(N, layout.C, dtype=torch.float64, device=code.device)
    check(lib().ddg_decode_table(_ptr(code, torch.float32), _ptr(m8, torch.uint8), _ptr(obs, torch.float64), _ptr(off),
                                 _ptr(cat), _ptr(dec_K, torch.float64), _ptr(dec_min, torch.float64),
                                 _ptr(dec_f32, torch.uint8), _ptr(pos_value, torch.float64), _ptr(out), N, layout.C,
                                 layout.D, _stream()), "ddg_decode_table")
    return out


# ----------------------------------------------------------------------------- 8(f) row 4: FD classifiers
class FdParams:
    """Device tensors of an FDModel's used layers (fc1, fc3: model/FD_model.py:15-26), updated in place by
    ddg_fd_train_epoch."""

    def __init__(self, W1, b1, W3, b3):
        self.W1, self.b1, self.W3, self.b3 = [t.detach().to(torch.float32).contiguous() for t in (W1, b1, W3, b3)]
        self.in_dim, self.out_dim = self.W1.shape[0], self.W3.shape[0]
        if self.W1.shape != (self.in_dim, self.in_dim) or self.W3.shape[1] != self.in_dim:
            raise _lib.DdgError("FdParams: fc1 must be [in,in] and fc3 [out,in]")
        self.c = _lib.FdWeights(self.in_dim, self.out_dim, self.W1.data_ptr(), self.b1.data_ptr(), self.W3.data_ptr(),
                                self.b3.data_ptr())


def fd_train_epoch(P: FdParams, X, Y, perm, batch: int = 256, lr: float = 0.02):
    """one epoch of train_Model (model/FD_model.py:41-46) over the rows of X in the order ``perm`` (int64, device)"""
    n = perm.numel()
    ws = _workspace(lib().ddg_fd_workspace_bytes(batch, P.in_dim, P.out_dim), X.device, "fd")
    check(lib().ddg_fd_train_epoch(C.byref(P.c), _ptr(X, torch.float32), _ptr(Y, torch.int32), _ptr(perm, torch.int64), n,
                                   int(batch), float(lr), ws.data_ptr(), ws.numel(), _stream()), "ddg_fd_train_epoch")


def fd_eval(P: FdParams, X, Y, chunk_rows: int = 65536):
    """-> (correct: 0-dim int32 device tensor, ce_mean: 0-dim float device tensor) of the model on (X, Y)"""
    n = X.shape[0]
    rows = min(n, chunk_rows)
    ws = _workspace(lib().ddg_fd_workspace_bytes(rows, P.in_dim, P.out_dim), X.device, "fd")
    hit = torch.empty(1, dtype=torch.int32, device=X.device)
    ce = torch.empty(1, dtype=torch.float32, device=X.device)
    check(lib().ddg_fd_eval(C.byref(P.c), _ptr(X, torch.float32), _ptr(Y, torch.int32), n, _ptr(hit), _ptr(ce),
                            ws.data_ptr(), ws.numel(), rows, _stream()), "ddg_fd_eval")
    return hit[0], ce[0]
