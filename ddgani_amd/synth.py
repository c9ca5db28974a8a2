"""Synthetic tables of the BASELINE.json shapes, generated with torch on the
target device.  Missingness generators restate the reference's (utils/data_loader.py:16-21 MCAR,
utils/util.py:701-729 MAR, :766-786 Region; MAR/Region ignore the requested rate like the reference)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import torch

from .ops import FieldLayout

SEED = 20240807

WORKLOADS = {
    # name: (rows, columns, categorical every n-th column (0 = none), cardinality, miss type, rate, k, T)
    "c3": dict(rows=1_000_000, cols=64, cat_every=4, card=8, miss="MAR", rate=0.3, k=10, T=6,
               desc="synthetic 1M rows x 64 mixed cols (every 4th categorical, cardinality 8), MAR, top_k=10"),
    "c4": dict(rows=10_000_000, cols=128, cat_every=0, card=0, miss="MCAR", rate=0.5, k=10, T=6,
               desc="synthetic 10M rows x 128 numeric cols, MCAR 0.5, top_k=10"),
    # c5 is all numeric: the hot path (fill + sampler) is what it measures.  The FD plug-in of BASELINE's configs[4]
    # discovers nothing on such a table (model/FD_model.py:208-210 skips a numeric right-hand side, as the reference does);
    # it is exercised on mixed tables instead: hospital, adult, and 200 000 synthetic rows (tests/test_gpu_fd.py).
    "c5": dict(rows=2_000_000, cols=256, cat_every=0, card=0, miss="Region", rate=0.2, k=10, T=6,
               desc="synthetic 2M rows x 256 numeric cols, Region missingness, top_k=10"),
}


@dataclass
class Table:
    layout: FieldLayout
    X_attn: torch.Tensor      # [N,D] z-scored / one-hot code of the mean-prefilled table, times M (= X0)
    x_diff: torch.Tensor      # [N,D] min-max / one-hot code fed to the diffusion sampler
    M: torch.Tensor           # [N,D] fp32 expanded mask
    m8: torch.Tensor          # [N,C] uint8 mask
    W: torch.Tensor           # [C,D] column weights
    k: int
    T: int


def make_mask(kind: str, rate: float, N: int, C: int, numeric_cols: List[int], vals: torch.Tensor,
              g: torch.Generator, device) -> torch.Tensor:
    if kind == "MCAR":
        return (torch.rand(N, C, generator=g, device=device) < 1 - rate).to(torch.uint8)
    if kind == "MAR":
        m = torch.ones(N, C, dtype=torch.uint8, device=device)
        pick = numeric_cols[int(torch.randint(0, len(numeric_cols), (1,), generator=g, device=device).item())]
        order = torch.argsort(vals[:, pick])
        cnt = int(N * 0.4)
        start = int(torch.randint(0, N - cnt + 1, (1,), generator=g, device=device).item())
        rows = order[start:start + cnt]
        drop = torch.rand(cnt, C, generator=g, device=device) < 0.5
        drop[:, pick] = False
        m[rows] = (~drop).to(torch.uint8)
        return m
    if kind == "Region":
        m = torch.ones(N, C, dtype=torch.uint8, device=device)
        missing = int(N * C * 0.2)
        max_rows, max_cols = missing, 1
        while max_rows > N:
            max_rows //= 2
            max_cols = missing // max_rows
        r0 = int(torch.randint(0, N - max_rows + 1, (1,), generator=g, device=device).item())
        c0 = int(torch.randint(0, C - max_cols + 1, (1,), generator=g, device=device).item())
        m[r0:r0 + max_rows, c0:c0 + max_cols] = 0
        return m
    raise ValueError(kind)


def make_table(name: str, device, rows: int = 0) -> Table:
    w = WORKLOADS[name]
    N = rows or w["rows"]
    C = w["cols"]
    g = torch.Generator(device=device)
    g.manual_seed(SEED)
    widths = [w["card"] if (w["cat_every"] and (c % w["cat_every"] == w["cat_every"] - 1)) else 1 for c in range(C)]
    lay = FieldLayout(widths, device)
    numeric_cols = [c for c in range(C) if widths[c] == 1]
    vals = torch.rand(N, C, generator=g, device=device)            # numeric values U(0,1); cats derived below
    m8 = make_mask(w["miss"], w["rate"], N, C, numeric_cols, vals, g, device)
    X_attn = torch.zeros(N, lay.D, device=device)
    x_diff = torch.zeros(N, lay.D, device=device)
    for c in range(C):
        a, b = lay.offsets[c], lay.offsets[c + 1]
        obs = m8[:, c].bool()
        if widths[c] == 1:
            v = vals[:, c]
            mean_fill = (v * obs).sum() / N                         # Mean.py:15-19 divides by the TOTAL row count
            v = torch.where(obs, v, mean_fill)
            X_attn[:, a] = (v - v.mean()) / v.std(unbiased=False)   # field.py:41-42,67
            x_diff[:, a] = (v - v.min()) / (v.max() - v.min())      # field.py:36-39,65
        else:
            code = (vals[:, c] * widths[c]).long().clamp_(0, widths[c] - 1)
            mode = torch.bincount(code[obs], minlength=widths[c]).argmax()
            code = torch.where(obs, code, mode)                     # mode fill (Mean.py:25-32)
            X_attn[torch.arange(N, device=device), a + code] = 1.0
            x_diff[torch.arange(N, device=device), a + code] = 1.0
    M = torch.repeat_interleave(m8.float(), torch.tensor(widths, device=device), dim=1).contiguous()
    X_attn = (X_attn * M).contiguous()
    Wc = torch.rand(C, C, generator=g, device=device) * 1e-3
    Wc = (Wc + Wc.T) / 2
    Wc.fill_diagonal_(0.0)
    W = torch.repeat_interleave(Wc, torch.tensor(widths, device=device), dim=1).contiguous()
    return Table(lay, X_attn, x_diff.contiguous(), M, m8.contiguous(), W, w["k"], w["T"])


def random_generator_state(D: int, T: int, seed: int = 3407):
    """Random-init Generator_x0 weights (nn.Linear / nn.Embedding / BatchNorm default initialisers)."""
    import torch.nn as nn
    torch.manual_seed(seed)
    mods = {"linears.0": nn.Linear(D, 2 * D), "linears.1": nn.BatchNorm1d(2 * D), "linears.4": nn.Linear(2 * D, D),
            "linears.5": nn.BatchNorm1d(D), "linears.8": nn.Linear(D, 2 * D), "linears.9": nn.BatchNorm1d(2 * D),
            "linears.12": nn.Linear(2 * D, D), "time_embedding": nn.Embedding(T + 1, D)}
    sd = {}
    for k, m in mods.items():
        sd[k + ".weight"] = m.weight.detach()
        if getattr(m, "bias", None) is not None:
            sd[k + ".bias"] = m.bias.detach()
    return sd
