"""Host-side mirror of the reference's ``model/Diffusion.py`` sampler interface, backed by the
HIP kernels of libddgani_hip.so.

Same names, argument meaning and return values as the reference functions cited in each
docstring (file:line relative to the reference tree), so a user script written against
``model.Diffusion`` runs against this module.  The modules here are parameter containers
with the reference's ``state_dict`` key layout; all arithmetic happens in the HIP library.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
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


def _uniform_t(timestep) -> Optional[int]:
    if isinstance(timestep, int):
        return timestep
    if isinstance(timestep, torch.Tensor) and timestep.numel() > 0:
        # the sampler passes a constant vector (model/Diffusion.py:432); training passes mixed t
        return None
    return None


def forward_diffusion(DS: Diffusion_setting, x_0: torch.Tensor, timestep, seed: int = 0, stream_id: int = 0,
                      eps: Optional[torch.Tensor] = None):
    """reference: model/Diffusion.py:391-396 -> (x_t, eps).  ``timestep`` may be an int (all rows) or
    an int64 tensor [rows].  eps=None draws the noise on chip (Philox)."""
    return ops.forward_noise(x_0, timestep, DS.time_steps, eps=eps, seed=seed, stream_id=stream_id,
                             sched_dev=DS.table, return_eps=True)


def get_xt(DS, x_0, timestep, seed: int = 0, stream_id: int = 0, eps=None):
    """reference: model/Diffusion.py:398-403."""
    return ops.forward_noise(x_0, timestep, DS.time_steps, eps=eps, seed=seed, stream_id=stream_id,
                             sched_dev=DS.table)


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
        self.seed = 0
        self._calls = 0

    def device_params(self, refresh: bool = False) -> GenParams:
        dev = self.time_embedding.weight.device
        if self._params is None or refresh:
            self._params = GenParams(self.state_dict(), dev)
        return self._params

    def forward(self, x, t, drops=None):
        P = self.device_params(refresh=True)
        self._calls += 1
        return ops.denoiser_fwd(P, x, t if isinstance(t, int) else t.contiguous(), self.layout, drops=drops,
                                drop_p=0.2 if self.training else 0.0, seed=self.seed, stream_id=self._calls)


# ------------------------------------------------------------------------------------------
def G_D_onestep(generator_x0, DS, x_t, timestep, M, x, z=None, drops=None, seed: int = 0, stream_id: int = 0):
    """reference: model/Diffusion.py:412-424.  ``generator_x0`` is a Generator_x0 (this module) or a
    GenParams; ``timestep`` an int or a constant int64 vector as the reference passes it."""
    t = int(timestep[0].item()) if isinstance(timestep, torch.Tensor) else int(timestep)
    if isinstance(generator_x0, GenParams):
        raise TypeError("pass a Generator_x0 module, or use reverse_sample() with GenParams")
    pred = generator_x0(x_t, t, drops=drops)
    return ops.reverse_step(x_t, x, M, pred, t, DS.time_steps, z=z, seed=seed, stream_id=stream_id,
                            sched_host=DS.host)


def reverse_sample(P: GenParams, T: int, x: torch.Tensor, M: torch.Tensor, layout: FieldLayout,
                   randomness: Optional[ExplicitRandomness] = None, seed: int = 0, drop_p: float = 0.2,
                   sched_host: Optional[np.ndarray] = None) -> torch.Tensor:
    """The T-step reverse diffusion of reverse_diffusion_G_D_sample (model/Diffusion.py:426-450):
    x_T = q(x, T); t = T..3: one generator call + fused reverse step; t = 2: x0_hat = G(x_2, 2).
    Throughput mode (randomness=None) draws all noise / dropout on chip, keyed by ``seed``."""
    sh = ops.schedule(T) if sched_host is None else sched_host
    r = randomness
    x_t = ops.forward_noise(x, T, T, eps=None if r is None else r.eps_T, seed=seed, stream_id=1)
    pred = torch.empty_like(x)
    i = 0
    for ts in range(T, 1, -1):
        ops.denoiser_fwd(P, x_t, ts, layout, drops=None if r is None else r.drops[i], drop_p=drop_p, seed=seed,
                         stream_id=100 + ts, out=pred)
        if ts > 2:
            ops.reverse_step(x_t, x, M, pred, ts, T, z=None if r is None else r.zs[i], seed=seed,
                             stream_id=200 + ts, out=x_t, sched_host=sh)
        i += 1
    return pred


def reverse_diffusion_G_D_sample(generator_x, discriminator_noise_x, DS, timesteps, no, dim, device, x, M_tensor,
                                 randomness: Optional[ExplicitRandomness] = None, seed: int = 0,
                                 eps2: Optional[torch.Tensor] = None):
    """reference: model/Diffusion.py:426-450 -> (x0_hat [N,D], discriminator_out [N,C]).
    The discriminator tail (:442-449) is dead in the reference's default flow but returned for API
    parity; it runs on the GPU here (the reference moves it to the CPU)."""
    P = generator_x.device_params(refresh=True)
    x0_hat = reverse_sample(P, timesteps, x, M_tensor, generator_x.layout, randomness=randomness, seed=seed,
                            drop_p=0.2 if generator_x.training else 0.0, sched_host=DS.host)
    d_out = None
    if discriminator_noise_x is not None:
        from .Discriminator_model import sampler_disc_tail
        d_out = sampler_disc_tail(discriminator_noise_x, DS, x, x0_hat, eps2=eps2, seed=seed)
    return x0_hat, d_out
