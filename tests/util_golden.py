"""Helpers shared by the tests: fixture loading (plain npz, no pickle)."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def state_dict(z, prefix):
    """'G__linears__0__weight' -> {'linears.0.weight': tensor}"""
    out = {}
    for k in z.files:
        if k.startswith(prefix):
            out[k[len(prefix):].replace("__", ".")] = torch.tensor(z[k])
    return out


def layout_of(widths):
    offsets = [0]
    for w in widths:
        offsets.append(offsets[-1] + int(w))
    return offsets, [int(w) > 1 for w in widths]


def sampler_randomness(z, T):
    """explicit draws of a sampler fixture: zs (T-2 tensors), masks (T-1 lists of 3 float masks)"""
    zs = [torch.tensor(z[f"z{i}"]) for i in range(T - 2)]
    masks_b = [[z[f"mask{i}_{l}"] for l in range(3)] for i in range(T - 1)]
    masks_f = [[torch.tensor(m.astype(np.float32) / np.float32(0.8)) for m in ms] for ms in masks_b]
    return zs, masks_b, masks_f
