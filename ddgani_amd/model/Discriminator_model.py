"""Host-side mirror of the reference's ``model/Discriminator_model.py`` (time-conditioned
discriminator), backed by ddg_disc_fwd."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from ..ops import DiscParams


class D(nn.Module):
    """Plain discriminator of the reference (model/Discriminator_model.py:5-39): three Linear layers
    (input->input->input->out) with ReLU + Dropout(0.2) and a sigmoid; the two BatchNorm members are
    constructed (state_dict parity) but unused, as in the reference.  On the DDGANI path it is only
    constructed for its RNG draws (BaseLine/DiffGANI.py:52) and trained inside
    ``get_valid_data_index`` (utils/util.py:625-643) when fewer than 20 % of the rows are complete:
    a host-preparation step, kept in stock torch ops (SURVEY.md 8(f) "next")."""

    def __init__(self, input_dim, latent_dim, out_dim):
        super().__init__()
        self.input_dim, self.latent_dim, self.out_dim = input_dim, latent_dim, out_dim
        self.fc1 = nn.Linear(input_dim, input_dim)
        self.fc2 = nn.Linear(input_dim, input_dim)
        self.fc3 = nn.Linear(input_dim, out_dim)
        self.batch_normal1 = nn.BatchNorm1d(latent_dim)
        self.batch_normal2 = nn.BatchNorm1d(out_dim)
        self.dropout = nn.Dropout(0.2)

    def forward(self, new_x):
        out = self.dropout(torch.relu(self.fc1(new_x)))
        out = self.dropout(torch.relu(self.fc2(out)))
        return torch.sigmoid(self.fc3(out))


class Discriminator_noise_x(nn.Module):
    """Parameter container, reference layout (model/Discriminator_model.py:82-102); forward =
    cat -> 3x[Linear, + step_embedding[t], ReLU] -> Linear -> sigmoid (:105-115) on the GPU."""

    def __init__(self, input_dim, latent_dim, output_dim, n_steps):
        super().__init__()
        self.linears = nn.ModuleList([nn.Linear(input_dim, latent_dim), nn.ReLU(), nn.Linear(latent_dim, latent_dim),
                                      nn.ReLU(), nn.Linear(latent_dim, latent_dim), nn.ReLU(),
                                      nn.Linear(latent_dim, output_dim)])
        self.step_embeddings = nn.ModuleList([nn.Embedding(n_steps, latent_dim) for _ in range(3)])
        self._params = None

    def device_params(self, refresh: bool = False) -> DiscParams:
        if self._params is None or refresh:
            self._params = DiscParams(self.state_dict(), self.linears[0].weight.device)
        return self._params

    def forward(self, xtj1, xt, t):
        return ops.disc_fwd(self.device_params(refresh=True), xtj1, xt, t if isinstance(t, int) else t.contiguous())


def sampler_disc_tail(disc: Discriminator_noise_x, DS, x, x0_hat, eps2: Optional[torch.Tensor] = None,
                      seed: int = 0, stream_id: int = 300):
    """The t=2 tail of the reference sampler (model/Diffusion.py:442-447): X_2 = q(x,2;eps2);
    x_1 = posterior mean(x0_hat, X_2) + std*eps2 where the reference's ``std`` already contains eps2
    (so the noise enters squared, :443-445); D(x_1, X_2, 2).  A reverse step with M=0 and z=eps2^2
    is exactly that expression."""
    T = DS.time_steps
    X_t, e = ops.forward_noise(x, 2, T, eps=eps2, seed=seed, stream_id=stream_id, return_eps=True)
    xtj1 = ops.reverse_step(X_t, x, torch.zeros_like(x), x0_hat, 2, T, z=e * e, sched_host=DS.host)
    return disc(xtj1, X_t, 2)
