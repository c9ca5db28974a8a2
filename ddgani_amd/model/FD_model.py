"""Host-side mirror of the reference's ``model/FD_model.py`` ("data-dependency" plug-in, ``--UseFD True``): same
names, arguments and results for the functions on the DDGANI path --

  get_FD_model_Tree   (:113-176)  entropy-based discovery of exact functional dependencies  X -> y  between
                                  categorical columns (left-hand side of at most two columns), then one classifier
                                  per dependency;
  get_model_by_tree   (:179-209)  training rows of one dependency and its FDModel;
  train_Model / test_model (:33-65), get_my_FD_loss (:76-111), FDModel (:12-29), entropy (:67-73).

The discovery is re-written vectorised: the reference's per-group entropy loop (isCurNodeFD, :279-307) asks whether
every group of left-hand values sees a single right-hand value, which is a count of distinct (X) against distinct
(X, y) code tuples over the co-observed rows.  The classifiers train and evaluate on the GPU
(ddg_fd_train_epoch / ddg_fd_eval).  The FD-refinement machinery (:310-619) is dead code in the reference (its only
call is commented out, model/Diffusion.py:599-605) and is not built.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset

from .. import ops


class FDModel(nn.Module):
    """reference: model/FD_model.py:12-29.  Parameter container with the reference's members (fc1, fc2, fc3 drawn
    from torch's global RNG in that order; fc2 is never used in forward)."""

    def __init__(self, input_size, output_size, x_index_list, y_index):
        super().__init__()
        self.fc1 = nn.Linear(input_size, input_size)
        self.relu = nn.ReLU()
        self.fc2 = nn.Linear(input_size, input_size)
        self.fc3 = nn.Linear(input_size, output_size)
        self.x_index_list = x_index_list
        self.y_index = y_index
        self._params = None

    def device_params(self, refresh: bool = False) -> ops.FdParams:
        if self._params is None or refresh:
            self._params = ops.FdParams(self.fc1.weight, self.fc1.bias, self.fc3.weight, self.fc3.bias)
        return self._params

    def write_back(self):
        """trained device tensors -> the module's parameters (state_dict interchange with the reference)"""
        P = self._params
        with torch.no_grad():
            for dst, src in ((self.fc1.weight, P.W1), (self.fc1.bias, P.b1), (self.fc3.weight, P.W3), (self.fc3.bias, P.b3)):
                dst.copy_(src)

    def set(self, new_x_index_list):
        self.x_index_list = new_x_index_list


def test_model(x, y, model, perm_source=None):
    """reference: model/FD_model.py:55-65 -> accuracy.  The reference walks the shuffling DataLoader once more, which
    draws from torch's global generator; ``perm_source`` (the index loader of train_Model) keeps that side effect."""
    if perm_source is not None:
        for _ in perm_source:
            pass
    hit, _ = ops.fd_eval(model.device_params(), x, y)
    return int(hit.item()) / x.shape[0]


def train_Model(x, y, model):
    """reference: model/FD_model.py:33-52.  300 epochs of SGD(lr 0.02), batch 256, rows reshuffled per epoch by a
    DataLoader whose RandomSampler draws from torch's global generator -- an index DataLoader of the same class makes
    the same draws in the same order, so a run seeded like the reference walks the same batches; early exit when the
    training accuracy is 1 at epochs 0 / 100 / 200."""
    dev = x.device
    y32 = y.to(torch.int32).contiguous()
    x = x.contiguous()
    P = model.device_params(refresh=True)
    loader = DataLoader(TensorDataset(torch.arange(x.shape[0])), batch_size=256, shuffle=True)
    for epoch in range(300):
        perm = torch.cat([b[0] for b in loader]).to(dev)
        ops.fd_train_epoch(P, x, y32, perm, batch=256, lr=0.02)
        if epoch % 100 == 0:
            if test_model(x, y32, model, perm_source=loader) == 1:
                break
    test_model(x, y32, model, perm_source=loader)
    model.write_back()
    return model


def entropy(labels):
    """reference: model/FD_model.py:67-73."""
    if len(labels) == 0:
        return 0
    _, counts = np.unique(labels, return_counts=True)
    p = counts / len(labels)
    return -np.sum(p * np.log2(p))


def _col_dims(fields):
    dims = [0]
    for f in fields:
        dims.append(dims[-1] + (len(f.dict) if f.data_type == "Categorical Data" else 1))
    return dims


def _fd_input(code, fields, x_index, y_index):
    """the classifier's input: the one-hot blocks of the left-hand columns, zeros for every other column except the
    right-hand one, which is dropped (model/FD_model.py:95-103, :193-201)"""
    dims = _col_dims(fields)
    parts = []
    for c in range(len(fields)):
        blk = code[:, dims[c]:dims[c + 1]]
        if c in x_index:
            parts.append(blk)
        elif c != y_index:
            parts.append(torch.zeros_like(blk))
    yb = code[:, dims[y_index]:dims[y_index + 1]]
    return torch.cat(parts, dim=1).contiguous(), torch.argmax(yb, dim=1)


def get_my_FD_loss(FD_model_list, decode_code, fields):
    """reference: model/FD_model.py:76-111: sum over the dependencies of CrossEntropyLoss(model(x), arg-max of the
    right-hand block), evaluated without gradients: a constant in the generator loss."""
    if len(FD_model_list) == 0:
        return 0
    total = 0
    for m in FD_model_list:
        x, y = _fd_input(decode_code, fields, m.x_index_list, m.y_index)
        _, ce = ops.fd_eval(m.device_params(), x, y.to(torch.int32).contiguous())
        total = total + ce
    return total


# ------------------------------------------------------------------------------------------ discovery
def _encode_columns(miss_data, categorical_cols):
    """integer codes per column as the reference builds them (:131-152): categorical -> rank of the value as a
    string (+1); numeric -> index of the last of ten quantile chunks of the sorted column that holds the value"""
    data = miss_data.values
    out = []
    for ci in range(data.shape[1]):
        arr = np.array(data[:, ci])
        if ci in categorical_cols:
            _, enc = np.unique(arr.astype(np.str_), return_inverse=True)
            enc = enc + 1
        else:
            # chunk i of np.array_split(sorted, 10) that holds the value's LAST occurrence (the reference assigns
            # enc[isin(arr, chunk_i)] = i for i = 0..9 in turn: O(N^2 / 10) there, two binary searches per value here)
            v = arr.astype(np.float64)
            srt = np.sort(v)
            n = len(srt)
            sizes = np.full(10, n // 10)
            sizes[:n % 10] += 1
            starts = np.concatenate([[0], np.cumsum(sizes)[:-1]])
            last = np.searchsorted(srt, v, side="right") - 1
            enc = np.searchsorted(starts, last, side="right") - 1
            enc[np.isnan(v) | (last < 0)] = 0                       # NaN is in no chunk: stays 0 as in the reference
            enc = enc.astype(object) if arr.dtype == object else enc.astype(arr.dtype)
        out.append(enc)
    return out


def isCurNodeFD(encoder_data, col_index, observed, cur_node):
    """reference: model/FD_model.py:279-307.  True when, over the rows observed in the right-hand column and in
    every left-hand column, the conditional entropy H(y | X) is zero: every group of X values has one y value."""
    if len(cur_node) == 0:
        return False
    rows = observed[col_index]
    for node in cur_node:
        rows = rows & observed[node]
    if not rows.any():
        return False
    X = np.stack([np.asarray(encoder_data[n][rows]) for n in cur_node], axis=1)
    y = np.asarray(encoder_data[col_index][rows]).reshape(-1, 1)
    nx = len(np.unique(X, axis=0))
    nxy = len(np.unique(np.concatenate([X, y], axis=1), axis=0))
    return nx == nxy


def updateFeatLabels(new_node, featLabels):
    """reference: model/FD_model.py:259-268."""
    drop = [n for n in featLabels if set(new_node) <= set(n) and len(new_node) > 0]
    for n in drop:
        featLabels.remove(n)
    featLabels.append(new_node)


def isCurNodeInFDs(cur_node, featLabels):
    """reference: model/FD_model.py:270-276."""
    if len(featLabels) == 0 or len(cur_node) == 0:
        return False
    return any(set(n) <= set(cur_node) for n in featLabels)


def _dfs(encoder_data, col_index, observed, featLabels, n_rows, graph, cur_node, start_node):
    """reference: model/FD_model.py:237-257 (same visiting order, same early exits)."""
    cur_node.append(start_node)
    if isCurNodeInFDs(cur_node, featLabels):
        return
    if n_rows == 0:
        return
    if len(cur_node) == 3:
        return
    if isCurNodeFD(encoder_data, col_index, observed, cur_node):
        updateFeatLabels(cur_node.copy(), featLabels)
        return
    for node in graph[start_node]:
        if node not in cur_node:
            _dfs(encoder_data, col_index, observed, featLabels, n_rows, graph, cur_node, node)
            cur_node.pop()


def discover_FDs(miss_data, data_m, categorical_cols, sort_corr_dict):
    """The dependency search of get_FD_model_Tree (:114-164) -> list of (x_index list, y_index) in the reference's
    order: for every categorical column y, depth-first over the other categorical columns ordered by their
    association with y, left-hand sides of one or two columns."""
    data_m = np.asarray(data_m)
    observed = [data_m[:, c] == 1 for c in range(data_m.shape[1])]
    encoder_data = _encode_columns(miss_data, categorical_cols)
    fds = []
    for y in range(data_m.shape[1]):
        if y not in categorical_cols:
            continue
        order = [i for i in sort_corr_dict[y] if i in categorical_cols and i != y]
        graph = {node: order[order.index(node) + 1:] for node in order}
        feat = []
        n_rows = int(observed[y].sum())
        for node in graph.keys():
            _dfs(encoder_data, y, observed, feat, n_rows, graph, [], node)
        for x_index in feat:
            fds.append((x_index, y))
    return fds


def get_model_by_tree(x_index, y_index, zero_feed_data, has_data_index, fields, device):
    """reference: model/FD_model.py:179-209.  ``has_data_index[c]``: boolean row mask of column c's observed cells
    (the reference keeps index lists and intersects them as Python sets, whose iteration order is not specified; rows
    are taken in ascending order here)."""
    rows = has_data_index[y_index].copy()
    for x in x_index:
        rows &= has_data_index[x]
    idx = torch.as_tensor(np.nonzero(rows)[0], dtype=torch.long, device=zero_feed_data.device)
    X, Y = _fd_input(zero_feed_data[idx], fields, x_index, y_index)
    dims = _col_dims(fields)
    model = FDModel(X.shape[1], dims[y_index + 1] - dims[y_index], x_index, y_index).to(device)
    return train_Model(X.to(device), Y.to(device), model)


def get_FD_model_Tree(miss_data, data_m, categorical_cols, zero_feed_data, fields, device, sort_corr_dict):
    """reference: model/FD_model.py:113-176 -> list of trained FDModel, one per discovered dependency."""
    data_m = np.asarray(data_m)
    has = [data_m[:, c] == 1 for c in range(data_m.shape[1])]
    models = []
    for x_index, y_index in discover_FDs(miss_data, data_m, categorical_cols, sort_corr_dict):
        m = get_model_by_tree(x_index, y_index, zero_feed_data, has, fields, device)
        if m is not None:
            models.append(m)
    return models
