"""Mean / mode pre-fill with the reference's semantics (BaseLine/Mean.py:5-32), vectorised."""
import numpy as np
import pandas as pd


def mean_fill(data):
    """reference: BaseLine/Mean.py:13-22.  NaNs of a numeric column become (sum of the observed
    values) / (TOTAL number of rows): the reference divides by ``data.shape[0]``, not by the number
    of observed values.  The sum runs in row order like the reference's loop.  In place."""
    obs = ~pd.isna(data)
    total = 0
    for v in data[obs]:           # sequential sum = the reference's accumulation order
        total = total + v
    data[~obs] = total / data.shape[0]
    return data


def zhong_fill(data):
    """reference: BaseLine/Mean.py:24-31.  NaNs of a categorical column become its most frequent
    value (first entry of pandas' ``value_counts``)."""
    frame = pd.DataFrame(data)
    mode = frame.value_counts().index.tolist()[0][0]
    frame.fillna(mode, inplace=True)
    return np.squeeze(frame.values)


def fill_data_mean(miss_data, con_cols):
    """reference: BaseLine/Mean.py:5-11.  Fills the [N,C] object/float array IN PLACE, column by column."""
    for i in range(miss_data.shape[1]):
        miss_data[:, i] = mean_fill(miss_data[:, i]) if i in con_cols else zhong_fill(miss_data[:, i])
    return miss_data
