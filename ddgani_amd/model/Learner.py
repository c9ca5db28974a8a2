"""Mirror of the reference's ``model/Learner.py`` ("data-utility plug-in"): a one-layer softmax
classifier trained for ``epoch`` full-batch SGD steps on the imputed code, whose validation loss
enters the generator loss only as a constant (computed under no_grad, model/Learner.py:62-64).
SURVEY.md 8(f) row 1 ("next"): kept in stock torch ops on the tensors' device for now; it is
numerically inert for the imputation except that it RE-SEEDS torch's global RNG (:44), which is
reproduced here because it changes the noise stream of the following epochs."""
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
    """reference: model/Learner.py:43-67 -> (validation loss tensor, validation accuracy)."""
    torch.manual_seed(3407)
    net = L(x_train.size()[1], label_num).to(device)
    opt = optim.SGD(net.parameters(), lr=0.01)
    criterion = nn.CrossEntropyLoss()
    xt = x_train.detach()
    yt = y_train.squeeze().long()
    for _ in range(epoch):
        loss = criterion(net(xt), yt)
        opt.zero_grad()
        loss.backward()
        opt.step()
    with torch.no_grad():
        Y = net(x_valid)
        yv = y_valid.squeeze().long()
        val_loss = criterion(Y, yv)
        accuracy = (torch.max(Y.data, dim=1)[1] == yv).sum().item() / y_valid.size(0)
    return val_loss, accuracy
