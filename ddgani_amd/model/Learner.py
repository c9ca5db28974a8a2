"""Mirror of the reference's ``model/Learner.py`` ("data-utility plug-in"): a one-layer softmax
classifier trained for ``epoch`` full-batch SGD steps on the imputed code, whose validation loss
enters the generator loss only as a constant (computed under no_grad, model/Learner.py:62-64).
SURVEY.md 8(f) row 1: the SGD steps and the validation pass run in ``ddg_learner_train`` (csrc/learner.hip),
one fused pass over the training rows per step; the module below only reproduces the reference's parameter
construction, because it RE-SEEDS torch's global RNG (:44) and its initialisers consume that stream, which
changes the noise of the following epochs."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.optim as optim


class L(nn.Module):
    """reference: model/Learner.py:10-40.  Members constructed in the reference's order (their
    initialisers consume the RNG); forward uses fc2 + softmax only."""

    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.fc1 = nn.Linear(input_dim, input_dim)
        self.fc2 = nn.Linear(input_dim, output_dim)
        self.fc3 = nn.Linear(input_dim // 3, output_dim)
        self.normal = nn.BatchNorm1d(input_dim)
        self.output_dim = output_dim
        self.softmax = nn.Softmax(dim=1)
        self.dropout = nn.Dropout(0.5)

    def forward(self, new_x):
        return self.softmax(self.fc2(new_x))


def train_L_code(epoch, x_train, y_train, x_valid, y_valid, label_num, device):
    """reference: model/Learner.py:43-67 -> (validation loss tensor, validation accuracy).
    Only fc2 is used by the reference's forward (:36-40), so only its parameters are trained."""
    from .. import ops
    torch.manual_seed(3407)
    net = L(x_train.size()[1], label_num).to(device)
    W = net.fc2.weight.data.contiguous()
    b = net.fc2.bias.data.contiguous()
    val_loss, accuracy = ops.learner_train(x_train, y_train, x_valid, y_valid, W, b, epoch, lr=0.01)
    return val_loss, accuracy
