"""Host-side mirror of the reference's ``utils/data_loader.py``: CSV load, 80/20 split, shuffle,
missingness injection, 'Null'/0 fill, label handling.  Same names, arguments and return tuples."""
from __future__ import annotations

import numpy as np
import pandas as pd
from sklearn.model_selection import train_test_split

from . import util


def binary_sampler(p, rows, cols, miss_seed):
    """reference: utils/data_loader.py:16-21.  MCAR mask: 1 where U(0,1) < p, numpy seeded with ``miss_seed``."""
    np.random.seed(miss_seed)
    return 1 * (np.random.uniform(0., 1., size=[rows, cols]) < p)


def data_loader(file_name, miss_rate, categorical_cols, label, miss_type, miss_seed):
    """reference: utils/data_loader.py:25-67 -> (data_x, miss_data_x, data_m, label frame, true_data,
    true_label, con_cols, nan_data, test_data, all_data).  ``miss_rate`` is honoured for MCAR only."""
    data_x = pd.read_csv(file_name)
    all_data = data_x.copy().iloc[:, :-1]
    train_data, test_data = train_test_split(data_x, test_size=0.2, random_state=miss_seed)
    data_x = train_data.sample(frac=1).reset_index(drop=True)          # global numpy RNG, like the reference
    label_column = data_x.iloc[:, -1]
    if label != "False":
        data_x = data_x.iloc[:, :-1]
    no, dim = data_x.shape
    con_cols = [i for i in range(dim) if i not in categorical_cols]
    if miss_type == "MCAR":
        data_m = binary_sampler(1 - miss_rate, no, dim, miss_seed)
    elif miss_type == "MAR":
        data_m = util.MAR(data_x, con_cols, miss_seed)
    elif miss_type == "MNAR":
        data_m = util.MNAR(data_x, con_cols, categorical_cols, miss_seed)
    elif miss_type == "Region":
        data_m = util.Region(data_x, miss_seed)
    else:
        raise ValueError(f"unknown MissType {miss_type!r}")
    complete = np.nonzero(np.all(np.asarray(data_m) == 1, axis=1))[0].tolist()
    if len(complete) == 0:
        true_data = true_label = None
    else:
        true_label = label_column.loc[complete]
        true_data = data_x.loc[complete]
    sim = data_x.copy()
    sim[np.asarray(data_m) == 0] = np.nan
    nan_data = sim.values
    miss_data_x = sim.copy()
    for i in range(dim):
        miss_data_x.iloc[:, i] = miss_data_x.iloc[:, i].fillna("Null" if i in categorical_cols else 0)
    return (data_x, miss_data_x, data_m, pd.DataFrame(label_column), true_data, true_label, con_cols, nan_data,
            test_data, all_data)


def value_loader(file_name, continuous_cols, label):
    """reference: utils/data_loader.py:94-108 -> (all column names, numeric names, categorical names)."""
    data_x = pd.read_csv(file_name)
    if label != "False":
        data_x = data_x.iloc[:, :-1]
    values = data_x.columns.values.tolist()
    value_num = [v for i, v in enumerate(values) if i in continuous_cols]
    value_cat = [v for i, v in enumerate(values) if i not in continuous_cols]
    return values, value_num, value_cat


def set_label(file_name, label_file_path, label):
    """reference: utils/data_loader.py:112-129.  Writes ``label_data.csv`` with the last column renamed
    'label' and mapped to 0..n-1 in order of first appearance; returns (path, number of labels)."""
    new_file_path = label_file_path + "label_data.csv"
    df = pd.read_csv(file_name)
    if label == "False":
        df.to_csv(new_file_path, index=False)
        return new_file_path, 0
    label_column = df.iloc[:, -1]
    df = df.iloc[:, :-1]
    df["label"] = label_column
    uniq = df["label"].unique()
    df["label"] = df["label"].map({v: i for i, v in enumerate(uniq)})
    df.to_csv(new_file_path, index=False)
    return new_file_path, len(uniq)
