#!/usr/bin/env python3
"""Golden fixtures of SURVEY 8(f) row 1 (the Learner), produced by RUNNING THE REFERENCE's
``model/Learner.py::train_L_code`` in the build container (needs /root/reference; never run on the GPU box).
The oracle restatement is asserted against the reference before anything is written; fixtures hold inputs, the
reference's (val_loss, accuracy) and the oracle's final fc2 parameters (the reference does not return them)."""
import os
import sys

import numpy as np
import torch

R = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, R)
sys.path.insert(0, ROOT)
from model import Learner as ref_learner          # noqa: E402
from oracle import ddgani_oracle as orc           # noqa: E402


def gen(name, seed, n_train, n_valid, D, L, epoch):
    rng = np.random.default_rng(seed)
    centers = rng.normal(size=(L, D)) * 0.8
    def draw(n):
        y = rng.integers(0, L, n)
        x = (centers[y] + rng.normal(size=(n, D))).astype(np.float32)
        x[:, : D // 3] = rng.uniform(size=(n, D // 3)) < 0.3        # some one-hot-like columns
        return x, y.astype(np.int64)
    xt, yt = draw(n_train)
    xv, yv = draw(n_valid)
    T = torch.tensor
    ref_loss, ref_acc = ref_learner.train_L_code(epoch, T(xt), T(yt)[:, None], T(xv), T(yv)[:, None], L, "cpu")
    o_loss, o_acc, W, b = orc.learner_train(epoch, T(xt), T(yt), T(xv), T(yv), L)
    assert abs(float(ref_loss) - o_loss) < 2e-6, (float(ref_loss), o_loss)
    assert ref_acc == o_acc, (ref_acc, o_acc)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), x_train=xt, y_train=yt, x_valid=xv, y_valid=yv,
                        L=np.int64(L), epoch=np.int64(epoch), val_loss=np.float64(float(ref_loss)),
                        acc=np.float64(ref_acc), W=W.numpy(), b=b.numpy())
    print(name, "val_loss", float(ref_loss), "acc", ref_acc, "oracle", o_loss)


if __name__ == "__main__":
    gen("learner_small", 11, 700, 180, 23, 3, 1000)
    gen("learner_wide", 12, 1500, 400, 300, 7, 200)
    gen("learner_manyclass", 13, 900, 300, 40, 20, 300)
