"""Host-side mirror of the reference's ``model/Diffusion.py`` sampler interface, backed by the
HIP kernels of libddgani_hip.so.

Same names, argument meaning and return values as the reference functions cited in each
docstring (file:line relative to the reference tree), so a user script written against
``model.Diffusion`` runs against this module.  The modules here are parameter containers
with the reference's ``state_dict`` key layout; all arithmetic happens in the HIP library.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import weakref

import torch
import torch.nn as nn

from .. import ops
from ..ops import FieldLayout, GenParams, DiscParams


# ------------------------------------------------------------------------------------------
class Diffusion_setting:
    """Noise schedule (reference: model/Diffusion.py:86-108).  Vectors are computed by
    ``ddg_schedule`` and exposed under the reference's attribute names as tensors on ``device``."""

    def __init__(self, time_steps, shape, device):
        self.time_steps = time_steps
        self.shape = shape
        self.device = device
        self.host = ops.schedule(time_steps)                   # [8,T] fp32, host
        self.table = torch.from_numpy(self.host.copy()).to(device)   # device copy for per-row t
        names = ("betas", "alphas", "sqrt_alphas", "one_by_sqrt_alpha_s", "sqrt_beta_s", "alpha_cumulative",
                 "sqrt_alpha_cumulative", "sqrt_one_minus_alpha_cumulative")
        for i, n in enumerate(names):
            setattr(self, n, self.table[i])


def get(element: torch.Tensor, idxs: torch.Tensor):
    """reference: model/Diffusion.py:382-388 (timestep t reads element t-1)."""
    return element.gather(-1, idxs - 1).reshape(-1, 1)


@dataclass
class ExplicitRandomness:
    """Every random draw of one sampler pass made explicit (parity mode):
    eps_T = noise of x_T; zs[i] = noise of the reverse step at t=T-i (t=T..3);
    drops[i] = three uint8 keep-masks of the generator call at t=T-i (t=T..2)."""
    eps_T: torch.Tensor
    zs: List[torch.Tensor]
    drops: List[Sequence[torch.Tensor]]


# ------------------------------------------------------------------------------------------
# On-chip random streams.  A Philox stream is (seed, stream_id); the kernels index elements inside it.
# The reference draws fresh numbers from torch's global generator on every call (randn_like, Dropout),
# so every call here gets a stream of its own when the caller names none:
#     stream_id = call << 24 | kind << 20 | rank << 12 | t          (t < 4096, rank < 256, kind < 16)
# with ``call`` a process-wide monotone counter, ``kind`` what the numbers are for, ``rank`` the shard
# and ``t`` the timestep.  Dropout (uniform draws) uses a salted SEED as well, so its raw Philox words
# never coincide with those of a normal stream; ddg_denoiser_fwd appends the layer as stream_id*4 + l.
# The default seed is torch.initial_seed(): torch.manual_seed() makes a run reproducible, as it does
# for the reference.
# ------------------------------------------------------------------------------------------
KIND_EPS, KIND_Z, KIND_EPS2, KIND_DROP = 1, 2, 3, 4
MAX_T = 4095
_DROP_SALT = 0x2545F4914F6CDD1D
_calls = itertools.count(1)


def next_call() -> int:
    return next(_calls)


def default_seed() -> int:
    return int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF


def stream_of(call: int, kind: int, t: int = 0, rank: int = 0) -> int:
    if not (0 <= t <= MAX_T and 0 <= rank < 256 and 0 < kind < 16 and 0 < call < (1 << 36)):
        raise ValueError(f"random stream out of range: call={call} kind={kind} rank={rank} t={t}")
    return (call << 24) | (kind << 20) | (rank << 12) | t


def drop_seed(seed: int) -> int:
    return (int(seed) ^ _DROP_SALT) & 0xFFFFFFFFFFFFFFFF


def _t_of(timestep) -> int:
    return int(timestep[0].item()) if isinstance(timestep, torch.Tensor) else int(timestep)


def forward_diffusion(DS: Diffusion_setting, x_0: torch.Tensor, timestep, seed: Optional[int] = None,
                      stream_id: Optional[int] = None, eps: Optional[torch.Tensor] = None):
    """reference: model/Diffusion.py:391-396 -> (x_t, eps).  ``timestep`` may be an int (all rows) or
    an int64 tensor [rows].  eps=None draws the noise on chip (Philox): fresh numbers on every call
    unless the caller names (seed, stream_id)."""
    seed = default_seed() if seed is None else seed
    stream_id = stream_of(next_call(), KIND_EPS) if stream_id is None else stream_id
    return ops.forward_noise(x_0, timestep, DS.time_steps, eps=eps, seed=seed, stream_id=stream_id,
                             sched_dev=DS.table, return_eps=True)


def get_xt(DS, x_0, timestep, seed: Optional[int] = None, stream_id: Optional[int] = None, eps=None):
    """reference: model/Diffusion.py:398-403."""
    return forward_diffusion(DS, x_0, timestep, seed=seed, stream_id=stream_id, eps=eps)[0]


class Generator_x0(nn.Module):
    """Parameter container with the reference's layout (model/Diffusion.py:145-178): same
    constructor signature, same ``state_dict`` keys, same default initialisers drawn in the same
    order from torch's global RNG, so weights initialise identically under the same seed.
    ``forward`` runs the HIP denoiser (train-mode BatchNorm + Dropout, model/Diffusion.py:179-201)."""

    def __init__(self, input_dim, encoder_dim, latent_dim, n_steps, fields):
        super().__init__()
        self.linears = nn.ModuleList([
            nn.Linear(input_dim, encoder_dim), nn.BatchNorm1d(encoder_dim), nn.ReLU(), nn.Dropout(0.2),
            nn.Linear(encoder_dim, latent_dim), nn.BatchNorm1d(latent_dim), nn.ReLU(), nn.Dropout(0.2),
            nn.Linear(latent_dim, encoder_dim), nn.BatchNorm1d(encoder_dim), nn.ReLU(), nn.Dropout(0.2),
            nn.Linear(encoder_dim, input_dim)])
        self.step_embeddings = nn.ModuleList([nn.Embedding(n_steps, encoder_dim), nn.Embedding(n_steps, latent_dim),
                                              nn.Embedding(n_steps, encoder_dim)])
        self.fields = fields
        self.time_embedding = nn.Embedding(n_steps, input_dim)
        self.layout = FieldLayout([f.dim() for f in fields])
        self._params = None

    def device_params(self, refresh: bool = False) -> GenParams:
        dev = self.time_embedding.weight.device
        if self._params is None or refresh:
            self._params = GenParams(self.state_dict(), dev)
        return self._params

    def forward(self, x, t, drops=None, seed: Optional[int] = None, stream_id: Optional[int] = None):
        """BatchNorm always normalises with the statistics of the batch at hand, as the reference's generator does
        in every call it ever makes (it is never put in eval mode, model/Diffusion.py:591,595):
        ``running_mean`` / ``running_var`` are not maintained.  ``eval()`` only switches Dropout off."""
        P = self.device_params(refresh=True)
        seed = default_seed() if seed is None else seed
        stream_id = stream_of(next_call(), KIND_DROP) if stream_id is None else stream_id
        return ops.denoiser_fwd(P, x, t if isinstance(t, int) else t.contiguous(), self.layout, drops=drops,
                                drop_p=0.2 if self.training else 0.0, seed=drop_seed(seed), stream_id=stream_id)


# ------------------------------------------------------------------------------------------
def G_D_onestep(generator_x0, DS, x_t, timestep, M, x, z=None, drops=None, seed: Optional[int] = None,
                call: Optional[int] = None):
    """reference: model/Diffusion.py:412-424.  ``generator_x0`` is a Generator_x0 (this module);
    ``timestep`` an int or a constant int64 vector as the reference passes it.  z / drops = None draw the
    noise and the dropout masks on chip, fresh on every call (``call`` = next_call() unless given)."""
    t = _t_of(timestep)
    if isinstance(generator_x0, GenParams):
        raise TypeError("pass a Generator_x0 module, or use reverse_sample() with GenParams")
    seed = default_seed() if seed is None else seed
    call = next_call() if call is None else call
    pred = generator_x0(x_t, t, drops=drops, seed=seed, stream_id=stream_of(call, KIND_DROP, t))
    return ops.reverse_step(x_t, x, M, pred, t, DS.time_steps, z=z, seed=seed, stream_id=stream_of(call, KIND_Z, t),
                            sched_host=DS.host)


def reverse_sample(P: GenParams, T: int, x: torch.Tensor, M: torch.Tensor, layout: FieldLayout,
                   randomness: Optional[ExplicitRandomness] = None, seed: Optional[int] = None, drop_p: float = 0.2,
                   sched_host: Optional[np.ndarray] = None, call: Optional[int] = None) -> torch.Tensor:
    """The T-step reverse diffusion of reverse_diffusion_G_D_sample (model/Diffusion.py:426-450):
    x_T = q(x, T); t = T..3: one generator call + fused reverse step; t = 2: x0_hat = G(x_2, 2).
    Throughput mode (randomness=None) draws all noise / dropout on chip: streams (seed, call, kind, t),
    ``call`` = next_call() unless given (so two passes never share numbers)."""
    sh = ops.schedule(T) if sched_host is None else sched_host
    r = randomness
    seed = default_seed() if seed is None else seed
    call = next_call() if call is None else call
    x_t = ops.forward_noise(x, T, T, eps=None if r is None else r.eps_T, seed=seed, stream_id=stream_of(call, KIND_EPS, T))
    pred = torch.empty_like(x)
    i = 0
    for ts in range(T, 1, -1):
        # t > 2: the head is fused into the reverse step (pre-head logits -> x_{t-1} in one pass); t = 2: x0_hat = head(y)
        ops.denoiser_fwd(P, x_t, ts, layout, drops=None if r is None else r.drops[i], drop_p=drop_p,
                         seed=drop_seed(seed), stream_id=stream_of(call, KIND_DROP, ts), out=pred, apply_head=ts == 2)
        if ts > 2:
            ops.head_reverse_step(pred, x_t, x, M, ts, T, layout, z=None if r is None else r.zs[i], seed=seed,
                                  stream_id=stream_of(call, KIND_Z, ts), out=x_t, sched_host=sh)
        i += 1
    return pred


def reverse_diffusion_G_D_sample(generator_x, discriminator_noise_x, DS, timesteps, no, dim, device, x, M_tensor,
                                 randomness: Optional[ExplicitRandomness] = None, seed: Optional[int] = None,
                                 eps2: Optional[torch.Tensor] = None):
    """reference: model/Diffusion.py:426-450 -> (x0_hat [N,D], discriminator_out [N,C]).
    The discriminator tail (:442-449) is dead in the reference's default flow but returned for API
    parity; it runs on the GPU here (the reference moves it to the CPU)."""
    P = generator_x.device_params(refresh=True)
    seed = default_seed() if seed is None else seed
    call = next_call()
    x0_hat = reverse_sample(P, timesteps, x, M_tensor, generator_x.layout, randomness=randomness, seed=seed,
                            drop_p=0.2 if generator_x.training else 0.0, sched_host=DS.host, call=call)
    d_out = None
    if discriminator_noise_x is not None:
        from .Discriminator_model import sampler_disc_tail
        d_out = sampler_disc_tail(discriminator_noise_x, DS, x, x0_hat, eps2=eps2, seed=seed,
                                  stream_id=stream_of(call, KIND_EPS2, 2))
    return x0_hat, d_out


# ==========================================================================================
# training side (model/Diffusion.py:245-379, :551-637)
# ==========================================================================================
class HostRandom:
    """Parity mode: every random draw of the reference's training / sampling loop is made on the
    host with the SAME generator calls in the SAME order as the reference's CPU run (numpy's global
    RNG for ``sample_x``, torch's global CPU generator for timesteps, noise and dropout masks:
    SURVEY.md appendix A.5) and uploaded.  A run seeded like the reference then sees the reference's
    random stream."""
    name = "host"

    def batch_indices(self, n_rows, batch):
        return list(np.random.choice(n_rows, batch, replace=False))          # utils/util.py:583

    def timesteps(self, T, batch, device):
        return torch.randint(low=2, high=T + 1, size=(batch,)).to(device)    # model/Diffusion.py:292

    def normal(self, rows, dim, device):
        return torch.randn_like(torch.empty(rows, dim)).to(device)           # model/Diffusion.py:392

    def dropout(self, rows, widths, device):
        # nn.Dropout(0.2) in train mode on the CPU draws empty_like(h).bernoulli_(0.8) per layer
        return [torch.empty(rows, w).bernoulli_(0.8).to(torch.uint8).to(device) for w in widths]

    def sampler_T(self, T, rows):
        torch.randint(low=T, high=T + 1, size=(rows,))                        # model/Diffusion.py:428 (value fixed)


class ChipRandom:
    """Throughput mode: noise and dropout are generated inside the kernels (Philox4x32-10 keyed by
    (seed, stream)); batch rows are drawn without replacement by a numpy Generator (O(batch))."""
    name = "chip"

    def __init__(self, seed=0):
        self.seed = int(seed)
        self.counter = 0
        self.gen = np.random.default_rng(seed)

    def next_stream(self):
        return next_call()

    def batch_indices(self, n_rows, batch):
        return self.gen.choice(n_rows, batch, replace=False, shuffle=False).tolist()

    def timesteps(self, T, batch, device):
        return torch.from_numpy(self.gen.integers(2, T + 1, size=batch)).to(device)


def consume_dead_module_rng(input_dim, n_steps):
    """The reference constructs an eps-MLP (``Diffusion``) and its EMA that are never trained or
    called (BaseLine/DiffGANI.py:49-51); their default initialisers advance torch's global RNG before
    the live networks are initialised.  Draw-equivalent throwaway layers keep the streams aligned:
    four Linear(D,D) and four Embedding(n_steps,D) (model/Diffusion.py:113-132)."""
    for _ in range(4):
        nn.Linear(input_dim, input_dim)
    for _ in range(4):
        nn.Embedding(n_steps, input_dim)


class _NetState:
    """Device parameters + gradient / Adam buffers of one network (the role of ``optim.Adam``)."""

    def __init__(self, params, keys, lr, weight_decay):
        self.P = params
        self.buf = ops.ParamGrads(params, keys)
        self.lr, self.wd = lr, weight_decay

    def step(self):
        self.buf.adam_step(lr=self.lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=self.wd)


def _train_batch(rnd, x, M, m_data, T, device, gen_widths, rank: int = 0):
    idx = rnd.batch_indices(x.shape[0], 128)                  # batch hard-coded to 128 (model/Diffusion.py:287)
    it = torch.as_tensor(idx, dtype=torch.long, device=device)
    x0, Mb, mb = x[it].contiguous(), M[it].contiguous(), m_data[it].contiguous()
    t = rnd.timesteps(T, x0.shape[0], device)
    if rnd.name == "host":
        eps = rnd.normal(x0.shape[0], x0.shape[1], device)
        drops = rnd.dropout(x0.shape[0], gen_widths, device)
        return x0, Mb, mb, t, dict(eps=eps, drops=drops)
    # one stream number for the step: the library derives eps (normal) and the three dropout layers from it
    return x0, Mb, mb, t, dict(seed=rnd.seed, stream_id=stream_of(rnd.next_stream(), KIND_EPS, rank=rank))


class GraphedSteps:
    """The generator and the discriminator minibatch step as HIP graphs (throughput mode, one process): a step is about
    45 short kernels whose launches, not their run time, set the pace.  Captured once per (network, loss scale): gather of
    the batch rows from a static index / timestep buffer, the gradient entry point, and the advance of a device counter
    that the library adds to every random-stream number (ops.set_stream_offset) -- a graph freezes kernel arguments, so
    the streams must move in device memory.  Per step the host draws the 128 row indices and timesteps into a ring of
    pinned buffers, copies the slot into the static buffer (stream-ordered), replays the graph and launches Adam (its bias
    corrections are host scalars).  A replay is bit-identical to the eager step with the same stream numbers, also over
    hundreds of unsynchronised steps (tests/test_gpu_graph_train.py).

    The word the library reads (``offset``) holds the replay counter only WHILE a step executes -- the first captured
    kernel copies the counter into it, the last one clears it -- so every other on-chip draw on the device (the sampler
    passes of the same run, another caller's (seed, stream_id)) sees an offset of zero and stays reproducible.  The
    library holds ONE such pointer per device: a second live instance is refused.

    Nothing inside the captured body may become a memset / memcpy NODE: the library zeroes and copies with kernels of its
    own (csrc/common.h::zero_async), the bookkeeping here uses arithmetic kernels (DESIGN.md section 5 has the defect that
    a ``hipMemsetAsync`` node caused)."""
    RING = 64
    BASE_STREAM = stream_of(1 << 35, KIND_EPS)      # call numbers of graph replays: 2^35 + 256 n, apart from next_call()
    _owner = None

    def __init__(self, x, M, m_data, T, device, batch=128):
        live = GraphedSteps._owner() if GraphedSteps._owner is not None else None
        if live is not None and live.counter is not None:
            raise RuntimeError("a GraphedSteps instance already holds the device's stream offset: close() it first")
        self.x, self.M, self.m = x, M, m_data
        self.T, self.B, self.dev = T, min(batch, x.shape[0]), device
        self.stage = torch.empty(self.RING, 2 * self.B, dtype=torch.int64).pin_memory()
        self.stage_np = self.stage.numpy()
        self.events = [None] * self.RING            # per slot: its last copy has run
        self.slot = 0
        self.feed = torch.zeros(2 * self.B, dtype=torch.int64, device=device)       # [indices | timesteps]
        self.feed[self.B:] = 2
        self.counter = torch.zeros(1, dtype=torch.int64, device=device)             # replays so far << 32
        self.offset = torch.zeros(1, dtype=torch.int64, device=device)              # what the library adds: 0 outside a step
        self.graphs = {}
        self.batch = (torch.empty((self.B, x.shape[1]), dtype=torch.float32, device=device),
                      torch.empty((self.B, x.shape[1]), dtype=torch.float32, device=device),
                      torch.empty((self.B, m_data.shape[1]), dtype=torch.float32, device=device))
        ops.set_stream_offset(self.offset)
        GraphedSteps._owner = weakref.ref(self)

    def close(self):
        """switches the stream offset off; the offset tensor must outlive every kernel that may read it"""
        if self.counter is None:
            return
        ref = GraphedSteps._owner
        if ref is not None and ref() in (self, None):
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream().synchronize()
            ops.set_stream_offset(None)
            GraphedSteps._owner = None
        self.counter = None
        self.graphs = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def same_data(self, x, M, m_data, T):
        return (x.data_ptr(), M.data_ptr(), m_data.data_ptr(), T) == (self.x.data_ptr(), self.M.data_ptr(),
                                                                       self.m.data_ptr(), self.T)

    def _body(self, fn):
        # one launch: counter -> offset, then the three gathers of the batch (kernels only: no memcpy / memset node in the
        # graph); one more after the step: counter += 2^32, offset = 0
        idx, t = self.feed[:self.B], self.feed[self.B:]
        x0, Mb, mb = ops.gather_batch(self.x, self.M, self.m, idx, out=self.batch, counter=self.counter, offset=self.offset)
        loss = fn(x0, Mb, mb, t)
        ops.step_end(self.counter, self.offset)
        return loss

    def step(self, key, fn, rnd):
        """fn(x0, Mb, mb, t) -> loss tensor, launching on the current stream with stream number BASE_STREAM."""
        ev = self.events[self.slot]
        if ev is not None:
            ev.synchronize()                                 # the slot's copy of the previous lap has run
        row = self.stage_np[self.slot]
        row[:self.B] = rnd.gen.choice(self.x.shape[0], self.B, replace=False, shuffle=False)
        row[self.B:] = rnd.gen.integers(2, self.T + 1, size=self.B)
        self.feed.copy_(self.stage[self.slot], non_blocking=True)
        if ev is None:
            ev = self.events[self.slot] = torch.cuda.Event()
        ev.record()
        self.slot = (self.slot + 1) % self.RING
        ent = self.graphs.get(key)
        if ent is None:
            first = self._body(fn)                           # eager: this step, and workspaces / caches exist before the capture
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss = self._body(fn)
            self.graphs[key] = (g, loss)
            return first
        ent[0].replay()
        return ent[1]


def _graph_steps(optimizer, rnd, comm, x, M, m_data, T, device, blocked=False):
    """the GraphedSteps of this training run, or None where the eager path applies (parity mode, several ranks, FD loss,
    DDG_TRAIN_GRAPH=0)."""
    import os
    if blocked or rnd.name != "chip" or (comm is not None and comm.on) or os.environ.get("DDG_TRAIN_GRAPH", "1") == "0":
        return None
    gs = getattr(rnd, "_graphs", None)
    if gs is None or not gs.same_data(x, M, m_data, T):
        if gs is not None:
            gs.close()
        gs = rnd._graphs = GraphedSteps(x, M, m_data, T, device)
    return gs


def train_one_epoch_Generator(model, discriminator_noise_x, DS, x, optimizer, loss_scaler, other_loss, epoch,
                              total_epochs, timesteps, device, field, M, m_data, loss_weight, FDs_model_list, fields,
                              zero_feed_data, rnd=None, comm=None):
    """reference: model/Diffusion.py:245-318.  ``timesteps`` minibatch steps of the generator:
    loss = Res*Res_weight + G_D*G_D_weight (+ the FD / Learner terms, constants under no_grad that do
    not reach a gradient).  ``optimizer`` is the generator's ``_NetState``; ``loss_scaler`` is unused
    (the reference's generator step does not use it either, :306-311).  Returns the mean loss.
    ``comm`` (ddgani_amd.dist.Comm of a multi-rank job): every rank draws the same 128 rows and timesteps, the
    step runs data-parallel over their rows (sync-BatchNorm + one gradient all-reduce over RCCL)."""
    rnd = rnd or HostRandom()
    PG, PD = optimizer.P, discriminator_noise_x.device_params()
    widths = (PG.enc_dim, PG.lat_dim, PG.enc_dim)
    total = 0.0
    losses, fd_losses = [], []
    dp = comm is not None and comm.on
    gs = _graph_steps(optimizer, rnd, comm, x, M, m_data, timesteps, device, blocked=len(FDs_model_list) != 0)
    for _ in range(timesteps):
        if gs is not None:
            def g_step(x0, Mb, mb, t):
                return ops.train_gen_grads(PG, PD, optimizer.buf, x0, Mb, mb, t, model.layout, DS.table, timesteps,
                                           loss_weight["Res_weight"], loss_weight["G_D_weight"], seed=rnd.seed,
                                           stream_id=GraphedSteps.BASE_STREAM)[0]
            l = gs.step(("G", id(optimizer), loss_weight["Res_weight"], loss_weight["G_D_weight"]), g_step, rnd)
            optimizer.step()
            losses.append(l.clone())
            continue
        x0, Mb, mb, t, kw = _train_batch(rnd, x, M, m_data, timesteps, device, widths, rank=comm.rank if dp else 0)
        if dp:
            from .. import dist as ddist
            l = ddist.sharded_train_step(comm, ops, "G", PG, PD, optimizer, x0, Mb, mb, t, model.layout, DS.table,
                                         timesteps, loss_weight=loss_weight, **kw)
        else:
            want_fd = len(FDs_model_list) != 0
            l, pred = ops.train_gen_grads(PG, PD, optimizer.buf, x0, Mb, mb, t, model.layout, DS.table, timesteps,
                                          loss_weight["Res_weight"], loss_weight["G_D_weight"], want_pred=want_fd, **kw)
            if want_fd:
                # Diffusion_FD_loss on zero_feed + (1-M) * Pred_x0 (:266-267): a constant under no_grad, reported only
                from .FD_model import get_my_FD_loss
                fd_losses.append(get_my_FD_loss(FDs_model_list, Mb * x0 + (1 - Mb) * pred, fields))
            optimizer.step()
        losses.append(l)
    if losses:
        ls = torch.stack(losses).double().cpu()
        total = float((ls[:, 0] * loss_weight["Res_weight"] + ls[:, 1] * loss_weight["G_D_weight"]).mean()) \
            + float(other_loss)
        if fd_losses:
            total += float(torch.stack(fd_losses).double().mean().cpu()) * loss_weight.get("FD_weight", 0)
    return total


def train_one_epoch_discriminator_noise_x(generator_x0, discriminator_noise_x, DS, x, optimizer, loss_scaler,
                                          other_loss, epoch, total_epochs, timesteps, device, field, M, m_data,
                                          loss_weight, rnd=None, comm=None):
    """reference: model/Diffusion.py:321-379.  ``timesteps`` minibatch steps of the discriminator on
    D_loss_func (scaled by D_weight on epochs divisible by 100, :338 vs :367).  ``comm``: as above."""
    rnd = rnd or HostRandom()
    PG, PD = generator_x0.device_params(), optimizer.P
    widths = (PG.enc_dim, PG.lat_dim, PG.enc_dim)
    scale = loss_weight["D_weight"] if epoch % 100 == 0 else 1.0
    losses = []
    dp = comm is not None and comm.on
    gs = _graph_steps(optimizer, rnd, comm, x, M, m_data, timesteps, device)
    for _ in range(timesteps):
        if gs is not None:
            def d_step(x0, Mb, mb, t):
                return ops.train_disc_grads(PG, PD, optimizer.buf, x0, Mb, mb, t, generator_x0.layout, DS.table,
                                            timesteps, loss_scale=scale, seed=rnd.seed,
                                            stream_id=GraphedSteps.BASE_STREAM)[0]
            l = gs.step(("D", id(optimizer), scale), d_step, rnd)
            optimizer.step()
            losses.append(l.clone())
            continue
        x0, Mb, mb, t, kw = _train_batch(rnd, x, M, m_data, timesteps, device, widths, rank=comm.rank if dp else 0)
        if dp:
            from .. import dist as ddist
            l = ddist.sharded_train_step(comm, ops, "D", PG, PD, optimizer, x0, Mb, mb, t, generator_x0.layout,
                                         DS.table, timesteps, loss_scale=scale, **kw)
        else:
            l, _ = ops.train_disc_grads(PG, PD, optimizer.buf, x0, Mb, mb, t, generator_x0.layout, DS.table,
                                        timesteps, loss_scale=scale, **kw)
            optimizer.step()
        losses.append(l)
    return float(torch.stack(losses).double().mean().cpu()) * scale if losses else 0.0


def host_sampler_randomness(rnd, T, rows, dim, widths, device):
    """The draws of one reverse_diffusion_G_D_sample call in the reference's order (A.5)."""
    rnd.sampler_T(T, rows)
    eps_T = rnd.normal(rows, dim, device)
    zs, drops = [], []
    for ts in range(T, 1, -1):
        if ts > 2:
            zs.append(rnd.normal(rows, dim, device))
            drops.append(rnd.dropout(rows, widths, device))
        else:
            drops.append(rnd.dropout(rows, widths, device))
            eps2 = rnd.normal(rows, dim, device)
    return ExplicitRandomness(eps_T=eps_T, zs=zs, drops=drops), eps2


def train_diffusion_discriminator(discriminator, generator_x, discriminator_noise_x, FDs_model_list, num_steps, epochs,
                                  lr, batch_size, loss_weight, data_m, impute_data_code, label_data, fields, value_cat,
                                  values, miss_data_x, enc, ori_data, continuous_cols, label_num, device, use_Learner,
                                  rng="host", sample_every=100, log=None, epoch_hook=None):
    """reference: model/Diffusion.py:551-637 -> imputed table (``.values`` of the best epoch's decode).
    ``lr`` and ``batch_size`` are accepted and ignored exactly like the reference (:571-573, :287).
    rng="host" replays the reference's CPU random streams (parity), rng="chip" draws on the GPU."""
    from ..utils.util import get_M_by_data_m, get_valid_data_index, reconvert_data, test_impute_data_rmse
    from .Learner import train_L_code
    import pandas as pd
    print("------------------Diffusion Discriminator train--------------------")
    impute_data = None
    torch.manual_seed(3047)
    dev = torch.device(device)
    generator_x.to(dev)
    discriminator_noise_x.to(dev)
    M_tensor = get_M_by_data_m(data_m, fields, dev)
    x = impute_data_code.to(dev).contiguous()
    zero_feed_data = M_tensor * x
    # validation split of the Learner (utils/util.py:623-646); its fallback discriminator trains on the device, with the
    # dropout keeps of the reference's CPU stream in parity mode
    valid_data_index = get_valid_data_index(data_m, discriminator, impute_data_code, dev, rng=rng, seed=3047)
    vset = set(int(i) for i in valid_data_index)
    train_data_index = [i for i in range(len(data_m)) if i not in vset]
    label_data_code = torch.FloatTensor(label_data.values).to(dev)
    vidx = torch.as_tensor(np.asarray(valid_data_index, dtype=np.int64), device=dev)
    tidx = torch.as_tensor(np.asarray(train_data_index, dtype=np.int64), device=dev)
    x_valid, y_valid = x[vidx], label_data_code[vidx]
    optG = _NetState(generator_x.device_params(refresh=True), ops.GEN_KEYS, 1e-4, 1e-4)
    optD = _NetState(discriminator_noise_x.device_params(refresh=True), ops.DISC_KEYS, 1e-4, 0.0)
    m_data = torch.tensor(np.asarray(data_m)).float().to(dev)
    no, dim = x.shape
    DS = Diffusion_setting(num_steps, dim, dev)
    rnd = HostRandom() if rng == "host" else ChipRandom(3047)
    widths = (optG.P.enc_dim, optG.P.lat_dim, optG.P.enc_dim)
    L_loss = 0
    r_M, b_e = 9999, 0
    # ARMSE / AMAE of every sampling pass on the device (one pass over the missing cells) instead of a pandas decode of
    # the whole table; the decode itself (reconvert_data) runs only for a pass that becomes the best one
    from ..utils.util import DeviceScorer
    try:
        scorer = DeviceScorer(fields, value_cat, values, ori_data, data_m, enc, continuous_cols, dev)
    except ValueError:
        scorer = None
    try:
        for epoch in range(1, epochs + 1):
            other_loss = L_loss * loss_weight["L_weight"]
            train_one_epoch_Generator(generator_x, discriminator_noise_x, DS, x, optG, None, other_loss, epoch, epochs,
                                      num_steps, dev, fields, M_tensor, m_data, loss_weight, FDs_model_list, fields,
                                      zero_feed_data, rnd=rnd)
            train_one_epoch_discriminator_noise_x(generator_x, discriminator_noise_x, DS, x, optD, None, other_loss, epoch,
                                                  epochs, num_steps, dev, fields, M_tensor, m_data, loss_weight, rnd=rnd)
            if epoch_hook is not None:
                epoch_hook(epoch, generator_x, discriminator_noise_x, valid_data_index)
            if epoch % sample_every == 0:
                if rng == "host":
                    rand, eps2 = host_sampler_randomness(rnd, num_steps, no, dim, widths, dev)
                    generator_data = reverse_sample(optG.P, num_steps, x, M_tensor, generator_x.layout, randomness=rand,
                                                    sched_host=DS.host)
                else:
                    generator_data = reverse_sample(optG.P, num_steps, x, M_tensor, generator_x.layout,
                                                    seed=rnd.seed, sched_host=DS.host)
                code = ops.mask_merge(x, M_tensor, generator_data, generator_x.layout)      # :598, :606-615
                if scorer is not None:
                    cur_r, cur_m = scorer.score(code)
                else:
                    cur_r, cur_m = test_impute_data_rmse(code, fields, value_cat, values, miss_data_x.copy(), data_m, enc,
                                                         ori_data, continuous_cols)
                if log is not None:
                    log.append((epoch, float(cur_r), float(cur_m)))
                if use_Learner == 'True':
                    L_loss, acc = train_L_code(1000, code[tidx], label_data_code[tidx], x_valid, y_valid, label_num, dev)
                    L_loss = float(L_loss)
                else:
                    L_loss = 0
                if cur_r < r_M:
                    if scorer is not None:       # (f3) decode on the device, labels mapped back on the host
                        impute_data = scorer.decode(code, miss_data_x.copy(), value_cat, values, enc)
                    else:
                        impute_data = pd.DataFrame(reconvert_data(code, fields, value_cat, values, miss_data_x.copy(),
                                                                  data_m, enc))
                    impute_data.columns = values
                    b_e, r_M = epoch, cur_r
    finally:
        gsteps = getattr(rnd, "_graphs", None)
        if gsteps is not None:       # graphed training steps: the library must not keep a pointer to their offset word
            gsteps.close()
            rnd._graphs = None
    print(b_e)
    return impute_data.values
