#!/usr/bin/env python3
"""BASELINE.json configs[0] at its FULL size: the reference itself on its ``adult`` table (30 148 rows, 24 118 after the
train split, 14 columns of which 8 categorical; MCAR 0.2, --AllSeed 42, all flags True, epochs overridden), numpy /
python / torch seeded to 0 right before ``data_loader``.  Build container only (imports /root/reference through
make_e2e's module set-up); peak host memory about 45 GB (the reference keeps all 14 N x N attention matrices).

Stored (plain arrays): data_m, the column-association map, the attention-filled table (ordinal codes), the reference's
ARMSE / AMAE per sampling pass and of its returned table, its validation split, weights after epochs 1 / 2 / 10.
adult.csv (one of the reference's data files = fixture data) is copied next to it.

    python tests/golden/make_e2e_adult.py [epochs=100] [table=adult] [miss_type=MCAR]
    python tests/golden/make_e2e_adult.py 100 spam MNAR      # BASELINE configs[1]: a numeric UCI table under MNAR (e2e_spam_MNAR.npz)
    DDG_REF_THREADS=1 DDG_SCORES_ONLY=1 python tests/golden/make_e2e_adult.py     # scores only (self-spread of the reference)
"""
import json
import os
import random
import shutil
import sys
import tempfile
import time
import types

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_e2e as m                                          # noqa: E402  (sys.path, torchmetrics stub, reference imports)


def main(epochs=100, name="adult", miss_type="MCAR", miss_rate=0.2, seed=42):
    threads = int(os.environ.get("DDG_REF_THREADS", "0"))
    if threads:
        torch.set_num_threads(threads)
    scores_only = os.environ.get("DDG_SCORES_ONLY") == "1"
    params = json.load(open(m.R + "/param/param.json"))
    param = dict([p for p in params if p["name"] == name][0])
    param["epochs"] = epochs
    work = tempfile.mkdtemp()
    os.chdir(work)
    path = "exp-dir/" + name + "/"
    os.makedirs(path, exist_ok=True)
    src_csv = m.R + "/" + param["file_path"]
    label_file, label_num = m.set_label(src_csv, path, "True")
    args = types.SimpleNamespace(UseAttention="True", UseLearner="True", UseFD="True", MissRate=miss_rate)
    log, snaps = [], {}
    orig = m.ref_diff.test_impute_data_rmse

    def spy(*a, **k):
        r = orig(*a, **k)
        log.append((float(r[0]), float(r[1])))
        return r
    m.ref_diff.test_impute_data_rmse = spy
    orig_d = m.ref_diff.train_one_epoch_discriminator_noise_x

    def spy_d(generator_x0, discriminator_noise_x, *a, **k):
        r = orig_d(generator_x0, discriminator_noise_x, *a, **k)
        epoch = a[5]
        if epoch in (1, 2, 10):
            snaps[f"G_b4_ep{epoch}"] = generator_x0.linears[12].bias.detach().numpy().copy()
            snaps[f"G_temb_ep{epoch}"] = generator_x0.time_embedding.weight.detach().numpy().copy()
            snaps[f"D_b6_ep{epoch}"] = discriminator_noise_x.linears[6].bias.detach().numpy().copy()
        return r
    m.ref_diff.train_one_epoch_discriminator_noise_x = spy_d
    orig_v = m.ref_diff.get_valid_data_index

    def spy_v(*a, **k):
        r = orig_v(*a, **k)
        snaps["valid_index"] = np.asarray(r, dtype=np.int64)
        return r
    m.ref_diff.get_valid_data_index = spy_v

    t0 = time.time()
    np.random.seed(0); random.seed(0); torch.manual_seed(0)
    (ori_data, miss_data, data_m, label_data, true_data, true_label, continuous_cols, nan_data, test_data,
     all_data) = m.data_loader(label_file, miss_rate, param["categorical_cols"], "True", miss_type, seed)
    values, value_num, value_cat = m.value_loader(label_file, continuous_cols, "True")
    miss_data.columns = values
    miss_data.to_csv(path + f"miss_data_{miss_rate}.csv", index=None)
    ori_code, enc = m.ref_util.categorical_to_code(all_data, value_cat, enc=None)
    corr = m.ref_attr.get_attr_map_Null_kendall(path + f"miss_data_{miss_rate}.csv", data_m, param["categorical_cols"],
                                                continuous_cols)
    print(f"loaded + column map: {time.time() - t0:.0f} s, rows {data_m.shape}", flush=True)
    fill = m.ref_imputer.get_Diff_acc_RMSE(nan_data, path, miss_rate, miss_data, enc, data_m, param["categorical_cols"],
                                           continuous_cols, value_cat, torch.device("cpu"), param, label_data, values,
                                           ori_data, label_num, args)
    fill_df = pd.DataFrame(fill, columns=values)
    armse, amae = m.ref_util.errorLoss(fill_df, ori_data, data_m, value_cat, continuous_cols, enc)
    print("reference:", name, "scores per pass", log, "final", armse, amae, f"{time.time() - t0:.0f} s", flush=True)
    if scores_only:
        print("SCORES_JSON " + json.dumps({"threads": torch.get_num_threads(), "scores": log, "final": [armse, amae]}))
        shutil.rmtree(work)
        return
    attn = pd.read_csv(path + f"Attention_Input_miss_data_{miss_rate}.csv")
    a = attn.copy()
    if value_cat:
        a[value_cat] = enc.transform(a[value_cat])
    shutil.copy(src_csv, os.path.join(HERE, name + ".csv"))
    tag = name if miss_type == "MCAR" else f"{name}_{miss_type}"
    np.savez_compressed(os.path.join(HERE, f"e2e_{tag}.npz"), miss_type=np.array(miss_type), scores=np.array(log),
                        armse=np.array(armse), amae=np.array(amae), data_m=np.asarray(data_m).astype(np.int8), corr=corr,
                        attn_coded=a.values.astype(float), epochs=np.array(epochs),
                        categorical_cols=np.array(param["categorical_cols"], dtype=np.int64), T=np.array(param["T"]),
                        top_k=np.array(param["top_k"]), miss_rate=np.array(miss_rate), seed=np.array(seed),
                        param_json=np.array(json.dumps(param)), **snaps)
    shutil.rmtree(work)


if __name__ == "__main__":
    main(epochs=int(sys.argv[1]) if len(sys.argv) > 1 else 100, name=sys.argv[2] if len(sys.argv) > 2 else "adult",
         miss_type=sys.argv[3] if len(sys.argv) > 3 else "MCAR")
