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
    # DDG_REF_THREADS=n DDG_SCORES_ONLY=1: run the reference with n intra-op threads and only print its scores as JSON
    # (make_e2e_spread.py: how far the reference drifts from ITSELF when only the fp32 summation order changes)
    threads = int(os.environ.get("DDG_REF_THREADS", "0"))
    if threads:
        torch.set_num_threads(threads)
    scores_only = os.environ.get("DDG_SCORES_ONLY") == "1"
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

    # every sampling pass (model/Diffusion.py:594-635): the networks' weights, the state of torch's CPU generator right
    # before the pass (its draws follow from it), the pass's outputs, the merged code and the decoded table.
    # For the first pass the draws themselves are stored too (tests check that the replay from the state reproduces them).
    passes = []
    orig_s = ref_diff.reverse_diffusion_G_D_sample

    def spy_s(generator_x, discriminator_noise_x, DS, timesteps, no, dim, device, x, M_tensor):
        rec = {"rng_state": torch.get_rng_state().numpy().copy(),
               "G": {k: v.detach().numpy().copy() for k, v in generator_x.state_dict().items()},
               "Dn": {k: v.detach().numpy().copy() for k, v in discriminator_noise_x.state_dict().items()}}
        if not passes:
            st = torch.get_rng_state()
            torch.randint(low=timesteps, high=timesteps + 1, size=(no,))
            rec["eps_T"] = torch.randn(no, dim).numpy()
            widths = (2 * dim, dim, 2 * dim)
            for ts in range(timesteps, 1, -1):
                if ts > 2:
                    rec[f"z_t{ts}"] = torch.randn(no, dim).numpy()
                for l, w_ in enumerate(widths):
                    rec[f"drop_t{ts}_{l}"] = np.packbits(torch.empty(no, w_).bernoulli_(0.8).numpy().astype(np.uint8), axis=1)
            rec["eps2"] = torch.randn(no, dim).numpy()
            torch.set_rng_state(st)
        r = orig_s(generator_x, discriminator_noise_x, DS, timesteps, no, dim, device, x, M_tensor)
        rec["x0_hat"] = r[0].detach().numpy().copy()
        rec["d_out"] = r[1].detach().numpy().copy()
        rec["x"] = x.detach().numpy().copy()
        passes.append(rec)
        return r
    ref_diff.reverse_diffusion_G_D_sample = spy_s

    def spy2(code, fields, value_cat_, values_, miss_x, data_m_, enc_, ori_, con_):
        miss_keep = miss_x.copy()            # reconvert_data ordinal-encodes its frame in place (utils/util.py:240)
        r = orig(code, fields, value_cat_, values_, miss_x, data_m_, enc_, ori_, con_)
        log.append((float(r[0]), float(r[1])))
        rec = passes[-1]
        rec["code"] = code.detach().numpy().copy()
        dec = pd.DataFrame(ref_util.reconvert_data(code, fields, value_cat_, values_, miss_keep, data_m_, enc_))
        dec.columns = values_
        if value_cat_:
            dec[value_cat_] = enc_.transform(dec[value_cat_])
        rec["decoded"] = dec.values.astype(float)
        rec["scores"] = np.array([float(r[0]), float(r[1])])
        return r
    ref_diff.test_impute_data_rmse = spy2

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
    if scores_only:
        dec = [p_["decoded"] for p_ in passes]
        print("SCORES_JSON " + json.dumps({"threads": torch.get_num_threads(), "scores": log, "final": [armse, amae],
                                           "decoded_sha": [__import__("hashlib").sha1(d.tobytes()).hexdigest() for d in dec],
                                           "G_b4_ep10_sum": float(np.abs(snaps["G_b4_ep10"]).sum()),
                                           "G_b4_ep100_sum": float(np.abs(snaps["G_b4_ep100"]).sum())}))
        np.save(os.path.join(work, "dec.npy"), np.stack(dec))
        shutil.copy(os.path.join(work, "dec.npy"), os.environ.get("DDG_DECODED_OUT", "/tmp/dec.npy"))
        np.save(os.environ.get("DDG_DECODED_OUT", "/tmp/dec.npy") + ".w.npy",
                np.concatenate([snaps[f"G_temb_ep{e}"].ravel() for e in (1, 2, 10, 100)]))
        shutil.rmtree(work)
        return
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
    flat = {"n_passes": np.array(len(passes))}
    for i, rec in enumerate(passes):
        for k_, v in rec.items():
            if isinstance(v, dict):
                for kk, vv in v.items():
                    flat[f"p{i}__{k_}__" + kk.replace(".", "__")] = vv
            else:
                flat[f"p{i}__{k_}"] = v
    np.savez_compressed(os.path.join(HERE, f"e2e_{tag}_passes.npz"), **flat)
    shutil.rmtree(work)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(name=a[0] if a else "hospital", epochs=int(a[1]) if len(a) > 1 else 200,
         miss_type=a[2] if len(a) > 2 else "MCAR", miss_rate=float(a[3]) if len(a) > 3 else 0.2)
