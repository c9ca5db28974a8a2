#!/usr/bin/env python3
"""Golden outputs of the reference's HOST functions (data_loader masks, column-association map,
mean/mode pre-fill, Data_convert, reconvert_data, errorLoss) on its own ``hospital`` table, produced
by importing and calling the reference (build container only).  Plain arrays, no pickle."""
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
tm.MeanMetric = type("MeanMetric", (), {})
sys.modules["torchmetrics"] = tm
import util as ref_util                                       # noqa: E402
from utils.data_loader import data_loader, value_loader, set_label   # noqa: E402
import get_attr_attn as ref_attr                              # noqa: E402
import Mean as ref_mean                                       # noqa: E402

cat_cols = [0, 1, 2, 3, 4, 5, 6]        # hospital: param/param.json categorical_cols
work = tempfile.mkdtemp() + "/"
label_file, label_num = set_label(os.path.join(HERE, "hospital.csv"), work, "True")
out = {"label_num": np.array(label_num)}
for mt in ("MCAR", "MAR", "MNAR", "Region"):
    np.random.seed(0); random.seed(0)
    r = data_loader(label_file, 0.3, cat_cols, "True", mt, 7)
    out[f"data_m_{mt}"] = np.asarray(r[2]).astype(np.int8)
    out[f"row0_{mt}"] = r[0].iloc[0].astype(str).values.astype("U")
np.random.seed(0); random.seed(0)
(ori, miss, data_m, label, _, _, con, nan_data, _, all_data) = data_loader(label_file, 0.2, cat_cols, "True", "MCAR", 42)
values, value_num, value_cat = value_loader(label_file, con, "True")
miss.columns = values
miss.to_csv(work + "miss.csv", index=None)
_, enc = ref_util.categorical_to_code(all_data, value_cat, enc=None)
out["corr"] = ref_attr.get_attr_map_Null_kendall(work + "miss.csv", data_m, cat_cols, con)
filled = ref_mean.fill_data_mean(nan_data.copy(), con)
fdf = pd.DataFrame(filled, columns=values)
code, _ = ref_util.categorical_to_code(fdf.copy(), value_cat, enc)
out["meanfill_coded"] = code.values.astype(float)
code.columns = list(range(code.shape[1]))
for model in ("minmax", "mean_std"):
    fields, feed = ref_util.Data_convert(code.copy(), model, con)
    out[f"feed_{model}"] = feed.values.astype(float)
    out[f"widths_{model}"] = np.array([f.dim() for f in fields])
# decode a perturbed code table and score it
rng = np.random.default_rng(1)
fields, feed = ref_util.Data_convert(code.copy(), "minmax", con)
x = torch.tensor(feed.values + rng.normal(size=feed.shape) * 0.3, dtype=torch.float)
dec = ref_util.reconvert_data(x, fields, value_cat, values, fdf.copy(), data_m, enc)
dec = pd.DataFrame(dec); dec.columns = values
armse, amae = ref_util.errorLoss(dec, ori, data_m, value_cat, con, enc)
dc = dec.copy(); dc[value_cat] = enc.transform(dc[value_cat])
out["decode_in"] = x.numpy()
out["decode_coded"] = dc.values.astype(float)
out["decode_armse"] = np.array(armse); out["decode_amae"] = np.array(amae)
out["M_expanded"] = ref_util.get_M_by_data_m(data_m, fields, torch.device("cpu")).numpy()
out["data_m"] = np.asarray(data_m).astype(np.int8)
np.savez_compressed(os.path.join(HERE, "host_hospital.npz"), **out)
print("wrote host_hospital.npz", {k: v.shape for k, v in out.items()})
