#!/usr/bin/env python3
"""BASELINE.json config 1 (UCI adult, MCAR 0.2, top_k from param.json = 0.05) as an attention-fill fixture, by RUNNING
THE REFERENCE on a row subsample of its `adult` table (the full table needs ~35 GB for the
reference's C dense N x N matrices; 2 500 sampled rows -> 2 000 after the reference's 80/20 split, k = 100).
Pipeline = the reference's own: set_label -> data_loader (MCAR 0.2) -> get_attr_map_Null_kendall -> mean / mode
pre-fill -> init_attn_2.  Stored like the other attention fixtures (X, W, data_m, offsets, k, new_code), so
tests/test_gpu_parity.py::test_attn_fill_golden and tests/test_oracle_golden.py pick it up.
Build container only (needs /root/reference)."""
import json
import os
import random
import sys
import tempfile

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402  (import recipe + reference modules + oracle)
from utils.data_loader import data_loader, value_loader, set_label   # noqa: E402
from BaseLine import Mean as ref_mean                      # noqa: E402
import get_attr_attn as ref_attr                           # noqa: E402

R = "/root/reference"


def main(rows=2500, miss_rate=0.2, seed=42):
    params = json.load(open(R + "/param/param.json"))
    param = dict([p for p in params if p["name"] == "adult"][0])
    work = tempfile.mkdtemp()
    os.chdir(work)
    os.makedirs("exp-dir/adult", exist_ok=True)
    path = "exp-dir/adult/"
    full = pd.read_csv(R + "/" + param["file_path"])
    sub = full.sample(n=rows, random_state=7).reset_index(drop=True)
    sub.to_csv("adult_sub.csv", index=None)
    label_file, label_num = set_label("adult_sub.csv", path, "True")
    np.random.seed(0); random.seed(0); torch.manual_seed(0)
    (ori_data, miss_data, data_m, label_data, _, _, continuous_cols, nan_data, _, all_data) = data_loader(
        label_file, miss_rate, param["categorical_cols"], "True", "MCAR", seed)
    values, _, value_cat = value_loader(label_file, continuous_cols, "True")
    miss_data.columns = values
    miss_data.to_csv(path + f"miss_data_{miss_rate}.csv", index=None)
    _, enc = mg.ref_util.categorical_to_code(all_data, value_cat, enc=None)
    corr = ref_attr.get_attr_map_Null_kendall(path + f"miss_data_{miss_rate}.csv", data_m, param["categorical_cols"],
                                              continuous_cols)
    filled = pd.DataFrame(ref_mean.fill_data_mean(nan_data, continuous_cols), columns=values)
    cat_cols = param["categorical_cols"]
    top_k = param["top_k"]
    out_df, new_code = mg.ref_util.init_attn_2(corr.copy(), filled.copy(), data_m.copy(), cat_cols, enc, value_cat,
                                               torch.device("cpu"), top_k)
    new_code = new_code.numpy()
    # the encoded operands, with the reference's own helpers (utils/util.py:398-407)
    code = filled.copy()
    for c in cat_cols:
        col = values[c]
        for val in code[col].value_counts().index.tolist():
            if val != "Null":
                code[col] = code[col].apply(lambda x: val if x == "Null" else x)
                break
    code, _ = mg.ref_util.categorical_to_code(code, value_cat, enc)
    C = code.shape[1]
    code.columns = list(range(C))
    fields, enc_code = mg.ref_util.Data_convert(code, "mean_std", continuous_cols)
    X = enc_code.values.astype(np.float32)
    offsets = [0]
    for f in fields:
        offsets.append(offsets[-1] + f.dim())
    W = mg.orc.expand_cols(corr, offsets).astype(np.float32)
    N = X.shape[0]
    k = int(N * top_k)
    dm = np.asarray(data_m).astype(np.int64)
    o = mg.orc.attn_fill_dense(torch.tensor(X), torch.tensor(W), dm, offsets, k).numpy()
    diff = np.abs(o - new_code).max()
    print(f"[attn_adult] N={N} D={X.shape[1]} k={k}: oracle dense vs reference max abs diff = {diff:.3e}")
    assert diff == 0.0, diff
    mg.save("attn_adult2000", data_m=dm, corr=corr, categorical_cols=np.array(cat_cols, dtype=np.int64),
            top_k=np.array(top_k), k=np.array(k), X=X, W=W, offsets=np.array(offsets, dtype=np.int64), new_code=new_code)


if __name__ == "__main__":
    main()
