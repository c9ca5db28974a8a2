#!/usr/bin/env python3
"""Golden vectors of the "data-dependency" plug-in (model/FD_model.py) by RUNNING THE REFERENCE on its `hospital`
table (MCAR 0.2, seed 42; same preparation as make_e2e.py).  Build container only (needs /root/reference).

Stored: the functional dependencies the reference's discovery finds (get_FD_model_Tree: for every FD its left-hand
columns and its right-hand column, in the reference's order), and for the first few FDs the classifier the reference
trains (initial weights, the training rows X / Y it builds, the weights after train_Model, with the state of torch's
CPU generator before the training so that the shuffles replay), plus the value of get_my_FD_loss on a decoded batch."""
import json
import os
import random
import sys
import tempfile
import types

import numpy as np
import pandas as pd
import torch

R = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
for p in (R, R + "/utils", R + "/BaseLine", R + "/param"):
    sys.path.insert(0, p)
tm = types.ModuleType("torchmetrics")
tm.MeanMetric = type("MeanMetric", (), {"__init__": lambda s: None, "update": lambda s, v: None,
                                        "compute": lambda s: torch.tensor(0.0)})
sys.modules["torchmetrics"] = tm

import util as ref_util                                       # noqa: E402
from utils.data_loader import data_loader, value_loader, set_label   # noqa: E402
from BaseLine import Mean as ref_mean                         # noqa: E402
from model import FD_model as ref_fd                          # noqa: E402
import get_attr_attn as ref_attr                              # noqa: E402

N_TRAINED = 3


def main(name="hospital", miss_rate=0.2, seed=42):
    params = json.load(open(R + "/param/param.json"))
    param = dict([p for p in params if p["name"] == name][0])
    work = tempfile.mkdtemp()
    os.chdir(work)
    os.makedirs("exp-dir/" + name, exist_ok=True)
    path = "exp-dir/" + name + "/"
    label_file, label_num = set_label(R + "/" + param["file_path"], path, "True")
    np.random.seed(0); random.seed(0); torch.manual_seed(0)
    (ori_data, miss_data, data_m, label_data, _, _, continuous_cols, nan_data, _, all_data) = data_loader(
        label_file, miss_rate, param["categorical_cols"], "True", "MCAR", seed)
    values, _, value_cat = value_loader(label_file, continuous_cols, "True")
    miss_data.columns = values
    miss_data.to_csv(path + f"miss_data_{miss_rate}.csv", index=None)
    _, enc = ref_util.categorical_to_code(all_data, value_cat, enc=None)
    corr = ref_attr.get_attr_map_Null_kendall(path + f"miss_data_{miss_rate}.csv", data_m, param["categorical_cols"],
                                              continuous_cols)
    sort_corr_dict = ref_util.sort_corr(corr)
    # the diffusion stage's input without the attention fill (mean / mode pre-fill only): what the FD stage sees is
    # M * code, so the pre-fill of the missing cells does not matter
    filled = pd.DataFrame(ref_mean.fill_data_mean(nan_data, continuous_cols), columns=values)
    coded, enc = ref_util.categorical_to_code(filled.copy(), value_cat, enc)
    coded.columns = list(range(coded.shape[1]))
    fields, feed = ref_util.Data_convert(coded, param["model_name"], continuous_cols)
    dev = torch.device("cpu")
    M = ref_util.get_M_by_data_m(data_m, fields, dev)
    x = torch.tensor(feed.values, dtype=torch.float)
    zero_feed = x * M

    fds = []
    trained = []
    orig = ref_fd.get_model_by_tree
    orig_train = ref_fd.train_Model

    def spy_train(X, Y, model):
        rec = {"X": X.numpy().copy(), "Y": Y.numpy().copy(),
               "W0": {k: v.detach().numpy().copy() for k, v in model.state_dict().items()},
               "rng_state": torch.get_rng_state().numpy().copy()}
        m = orig_train(X, Y, model)
        rec["W1"] = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
        trained.append(rec)
        return m

    def spy(x_index, y_index, zfd, has_data_index, fields_, device):
        fds.append((list(x_index), int(y_index)))
        if len(trained) < N_TRAINED:
            ref_fd.train_Model = spy_train
            try:
                return orig(x_index, y_index, zfd, has_data_index, fields_, device)
            finally:
                ref_fd.train_Model = orig_train
        return None
    ref_fd.get_model_by_tree = spy
    torch.manual_seed(1234)
    models = ref_fd.get_FD_model_Tree(filled, data_m, param["categorical_cols"], zero_feed, fields, dev, sort_corr_dict)
    print(len(fds), "functional dependencies;", len(models), "trained")
    # FD loss of the trained models on a "decoded batch" (any encoded rows: the first 128)
    batch = x[:128]
    loss = ref_fd.get_my_FD_loss(models, batch, fields)
    out = {"n_fd": np.array(len(fds)), "fd_y": np.array([y for _, y in fds], dtype=np.int64),
           "fd_x": np.array([xs + [-1] * (2 - len(xs)) for xs, _ in fds], dtype=np.int64),
           "data_m": np.asarray(data_m), "corr": corr, "filled_coded": coded.values.astype(float),
           "categorical_cols": np.array(param["categorical_cols"], dtype=np.int64), "x": x.numpy(),
           "widths": np.array([f.dim() for f in fields], dtype=np.int64),
           "fd_loss_batch128": np.array(float(loss)), "n_trained": np.array(len(trained))}
    for i, rec in enumerate(trained):
        for k, v in rec.items():
            if isinstance(v, dict):
                for kk, vv in v.items():
                    out[f"t{i}__{k}__" + kk.replace(".", "__")] = vv
            else:
                out[f"t{i}__{k}"] = v
    np.savez_compressed(os.path.join(HERE, "fd_hospital.npz"), **out)
    print("fd list", fds[:10], "... loss", float(loss))


if __name__ == "__main__":
    main()
