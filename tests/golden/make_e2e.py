#!/usr/bin/env python3
"""End-to-end golden values by RUNNING THE REFERENCE ITSELF on its own
``hospital`` data set (MCAR 0.2, --AllSeed 42, all flags True, epochs overridden) with numpy / python
/ torch seeded to 0 right before ``data_loader``.  Build container only (needs /root/reference).

Stored (plain arrays, no pickle): the ARMSE / AMAE the reference scores after every 100 epochs, its
final returned table (ordinal codes), data_m, the attention-filled table, the column-association map
and the input CSV's rows (hospital.csv is one of the reference's data files = fixture data)."""
import json
import os
import random
import shutil
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


class MeanMetric:  # logging accumulator only (model/Diffusion.py:247,281,317: value discarded)
    def __init__(self):
        self.s, self.n = 0.0, 0

    def update(self, v):
        self.s += float(v)
        self.n += 1

    def compute(self):
        return torch.tensor(self.s / max(self.n, 1))


tm.MeanMetric = MeanMetric
sys.modules["torchmetrics"] = tm

import util as ref_util                                       # noqa: E402
from utils.data_loader import data_loader, value_loader, set_label   # noqa: E402
from BaseLine import DiffGANI as ref_imputer                  # noqa: E402
from model import Diffusion as ref_diff                       # noqa: E402
import get_attr_attn as ref_attr                              # noqa: E402


def main(name="hospital", epochs=200, miss_type="MCAR", miss_rate=0.2, seed=42):
    params = json.load(open(R + "/param/param.json"))
    param = dict([p for p in params if p["name"] == name][0])
    param["epochs"] = epochs
    work = tempfile.mkdtemp()
    os.chdir(work)
    os.makedirs("exp-dir/" + name, exist_ok=True)
    path = "exp-dir/" + name + "/"
    src_csv = R + "/" + param["file_path"]
    label_file, label_num = set_label(src_csv, path, "True")
    args = types.SimpleNamespace(UseAttention="True", UseLearner="True", UseFD="True", MissRate=miss_rate)
    log = []
    orig = ref_diff.test_impute_data_rmse

    def spy(*a, **k):
        r = orig(*a, **k)
        log.append((float(r[0]), float(r[1])))
        return r
    ref_diff.test_impute_data_rmse = spy
    # weights after selected epochs (trajectory pins): spy on the discriminator epoch, the last call of an epoch
    snaps = {}
    orig_d = ref_diff.train_one_epoch_discriminator_noise_x

    def spy_d(generator_x0, discriminator_noise_x, *a, **k):
        r = orig_d(generator_x0, discriminator_noise_x, *a, **k)
        epoch = a[5]
        if epoch in (1, 2, 10, 100):
            snaps[f"G_b4_ep{epoch}"] = generator_x0.linears[12].bias.detach().numpy().copy()
            snaps[f"G_temb_ep{epoch}"] = generator_x0.time_embedding.weight.detach().numpy().copy()
            snaps[f"D_b6_ep{epoch}"] = discriminator_noise_x.linears[6].bias.detach().numpy().copy()
        return r
    ref_diff.train_one_epoch_discriminator_noise_x = spy_d
    orig_v = ref_diff.get_valid_data_index

    def spy_v(*a, **k):
        r = orig_v(*a, **k)
        snaps["valid_index"] = np.asarray(r, dtype=np.int64)
        return r
    ref_diff.get_valid_data_index = spy_v

    np.random.seed(0); random.seed(0); torch.manual_seed(0)
    (ori_data, miss_data, data_m, label_data, true_data, true_label, continuous_cols, nan_data, test_data,
     all_data) = data_loader(label_file, miss_rate, param["categorical_cols"], "True", miss_type, seed)
    values, value_num, value_cat = value_loader(label_file, continuous_cols, "True")
    miss_data.columns = values
    miss_data.to_csv(path + 'miss_data_{}.csv'.format(miss_rate), index=None)
    ori_code, enc = ref_util.categorical_to_code(all_data, value_cat, enc=None)
    corr = ref_attr.get_attr_map_Null_kendall(path + 'miss_data_{}.csv'.format(miss_rate), data_m,
                                              param["categorical_cols"], continuous_cols)
    fill = ref_imputer.get_Diff_acc_RMSE(nan_data, path, miss_rate, miss_data, enc, data_m, param["categorical_cols"],
                                         continuous_cols, value_cat, torch.device("cpu"), param, label_data, values,
                                         ori_data, label_num, args)
    fill_df = pd.DataFrame(fill, columns=values)
    armse, amae = ref_util.errorLoss(fill_df, ori_data, data_m, value_cat, continuous_cols, enc)
    print("reference:", name, "per-100-epoch scores", log, "final", armse, amae)
    attn = pd.read_csv(path + 'Attention_Input_miss_data_{}.csv'.format(miss_rate))

    def coded(df):
        d = df.copy()
        if value_cat:
            d[value_cat] = enc.transform(d[value_cat])
        return d.values.astype(float)
    shutil.copy(src_csv, os.path.join(HERE, name + ".csv"))
    tag = name if miss_type == "MCAR" else f"{name}_{miss_type}"
    np.savez_compressed(os.path.join(HERE, f"e2e_{tag}.npz"), miss_type=np.array(miss_type), scores=np.array(log), armse=np.array(armse),
                        amae=np.array(amae), data_m=np.asarray(data_m), corr=corr, fill_coded=coded(fill_df),
                        attn_coded=coded(attn), ori_coded=coded(ori_data), epochs=np.array(epochs),
                        categorical_cols=np.array(param["categorical_cols"], dtype=np.int64), T=np.array(param["T"]),
                        top_k=np.array(param["top_k"]), miss_rate=np.array(miss_rate), seed=np.array(seed),
                        param_json=np.array(json.dumps(param)), **snaps)
    shutil.rmtree(work)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(name=a[0] if a else "hospital", epochs=int(a[1]) if len(a) > 1 else 200,
         miss_type=a[2] if len(a) > 2 else "MCAR", miss_rate=float(a[3]) if len(a) > 3 else 0.2)
