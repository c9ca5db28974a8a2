#!/usr/bin/env python3
"""Drop-in driver with the reference's command line (test_main/main.py:64-181):

    python main.py --Data hospital --MissType MCAR --MissRate 0.2 --UseAttention True --UseLearner True --UseFD True

Run from a directory that holds ``param/param.json`` and the ``dataset/`` tree it names (the
reference's repository root); artefacts go to ``exp-dir/<name>/`` like the reference's.  Extra flags
(not in the reference): ``--Epochs`` overrides param.json's epoch count, ``--Repeats`` the 5 repeats,
``--Rng host|chip`` selects reference-stream parity (default) or on-chip random numbers."""
import argparse
import json
import os
import statistics
import sys

import pandas as pd
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ddgani_amd.BaseLine import DiffGANI                      # noqa: E402
from ddgani_amd.param.data_index import get_data_index       # noqa: E402
from ddgani_amd.utils import util                             # noqa: E402
from ddgani_amd.utils.data_loader import data_loader, set_label, value_loader   # noqa: E402


def run(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--config', help='a json config file', default='param/param.json')
    parser.add_argument('--Data', type=str, help='data_name', default='wine')
    parser.add_argument('--MissType', type=str, help='miss_type', default='MCAR')
    parser.add_argument('--MissRate', type=float, help='miss_rate', default='0.2')
    parser.add_argument('--AllSeed', type=int, help='miss_seed', default=42)
    parser.add_argument('--UseAttention', type=str, help='use or not use Attention', default='True')
    parser.add_argument('--UseLearner', type=str, help='use or not use Data-Utils plug-in', default='True')
    parser.add_argument('--UseFD', type=str, help='use or not use Data-Dependency plug-in', default='True')
    parser.add_argument('--Epochs', type=int, default=0)
    parser.add_argument('--Repeats', type=int, default=5)
    parser.add_argument('--Rng', type=str, default='host', choices=['host', 'chip'])
    args = parser.parse_args(argv)

    with open(args.config) as f:
        params = json.load(f)
    data_index = get_data_index(args.Data, params)
    if data_index == -1:
        raise SystemExit(f"dataset {args.Data!r} is not in {args.config}")
    param = dict(params[data_index])
    if args.Epochs:
        param["epochs"] = args.Epochs
    path = "exp-dir/" + param["name"] + "/"
    os.makedirs(path, exist_ok=True)
    label = "True"
    label_file, label_num = set_label(param["file_path"], path, label)
    categorical_cols = param['categorical_cols']
    if not torch.cuda.is_available():
        raise SystemExit("ddgani_amd needs an MI355X: there is no CPU path")
    device = torch.device('cuda:0')
    ARMSE_list, AMAE_list = [], []
    for _ in range(args.Repeats):
        miss_rate, miss_type, seed = float(args.MissRate), args.MissType, args.AllSeed
        (ori_data, miss_data, data_m, label_data, true_data, true_label, continuous_cols, nan_data, test_data,
         all_data) = data_loader(label_file, miss_rate, categorical_cols, label, miss_type, seed)
        values, value_num, value_cat = value_loader(label_file, continuous_cols, label)
        miss_data.columns = values
        miss_data.to_csv(path + 'miss_data_{}.csv'.format(miss_rate), index=None)
        ori_code, enc = util.categorical_to_code(all_data, value_cat, enc=None)
        fill_data_np = DiffGANI.get_Diff_acc_RMSE(nan_data, path, miss_rate, miss_data, enc, data_m, categorical_cols,
                                                  continuous_cols, value_cat, device, param, label_data, values,
                                                  ori_data, label_num, args, rng=args.Rng)
        fill_data = pd.DataFrame(fill_data_np, columns=values)
        ARMSE, AMAE = util.errorLoss(fill_data, ori_data, data_m, value_cat, continuous_cols, enc)
        print("dataset: {}, ARMSE: {:.4f}, AMAE: {:.4f}".format(param["name"], ARMSE, AMAE))
        ARMSE_list.append(ARMSE)
        AMAE_list.append(AMAE)
    ARMSE, AMAE = sum(ARMSE_list) / len(ARMSE_list), sum(AMAE_list) / len(AMAE_list)
    print("dataset: {}, missrate: {}, misstype: {}, ARMSE: {:.4f}±{:.4f}, AMAE: {:.4f}±{:.4f}, Use_A: {}, Use_L:{}, "
          "Use_FD:{}".format(param["name"], args.MissRate, args.MissType, ARMSE, statistics.pstdev(ARMSE_list), AMAE,
                             statistics.pstdev(AMAE_list), args.UseAttention, args.UseLearner, args.UseFD))
    return ARMSE_list, AMAE_list


if __name__ == "__main__":
    run()
