"""Column-association weights of the row attention, mirror of the reference's
``utils/get_attr_attn.py``: for every pair of table columns a (deliberately damped, ``+1e5``)
Cramer's-V statistic of the 20-bin discretised values over the rows observed in both columns.

Vectorised: the C^2 pandas crosstabs of the reference become integer-coded ``np.bincount``
contingency tables; the chi-square statistic is computed as ``scipy.stats.chi2_contingency`` does
(Pearson, Yates' continuity correction for 2x2 tables only)."""
from __future__ import annotations

import numpy as np
import pandas as pd


def _chi2(table: np.ndarray) -> float:
    """Pearson chi-square of a contingency table without empty rows / columns (what
    ``chi2_contingency(table)[0]`` returns: correction=True acts only when dof == 1)."""
    table = table.astype(np.float64)
    n = table.sum()
    expected = np.outer(table.sum(axis=1), table.sum(axis=0)) / n
    r, k = table.shape
    dof = (r - 1) * (k - 1)
    if dof == 0:
        return 0.0
    if dof == 1:
        diff = expected - table
        direction = np.sign(diff)
        magnitude = np.minimum(0.5, np.abs(diff))
        table = table + magnitude * direction
    return float(((table - expected) ** 2 / expected).sum())


def cramers_v_table(table: np.ndarray) -> float:
    """reference: utils/get_attr_attn.py:9-18 applied to a ready contingency table [r,k]."""
    chi2 = _chi2(table)
    n = table.sum()
    phi2 = chi2 / n
    r, k = table.shape
    phi2corr = max(0, phi2 - ((k - 1) * (r - 1)) / (n - 1))
    rcorr = r - ((r - 1) ** 2) / (n - 1)
    kcorr = k - ((k - 1) ** 2) / (n - 1)
    return np.sqrt(phi2corr / (min((kcorr - 1), (rcorr - 1)) + 1e5))


def cramers_v(x, y):
    """reference: utils/get_attr_attn.py:9-18 (x, y: equally long label sequences)."""
    _, xi = np.unique(np.asarray(x), return_inverse=True)
    _, yi = np.unique(np.asarray(y), return_inverse=True)
    ky = int(yi.max()) + 1 if len(yi) else 0
    table = np.bincount(xi * ky + yi, minlength=(int(xi.max()) + 1) * ky).reshape(-1, ky)
    return cramers_v_table(table)


def discretise(Miss_data: pd.DataFrame, continuous_cols) -> np.ndarray:
    """Integer codes [N,C]: numeric columns through ``pd.cut(bins=20)`` (the 0-filled missing cells
    take part in the range, as in the reference :49-51), categorical columns by sorted distinct value
    (what OrdinalEncoder does, :54).  Only the induced partition matters for the statistic."""
    n, C = Miss_data.shape
    codes = np.zeros((n, C), dtype=np.int64)
    for i in range(C):
        col = Miss_data.iloc[:, i]
        if i in continuous_cols:
            codes[:, i] = pd.cut(col, bins=20, labels=False).values
        else:
            _, inv = np.unique(col.values.astype(str) if col.dtype == object else col.values, return_inverse=True)
            codes[:, i] = inv
    return codes


def association_map(codes: np.ndarray, data_m: np.ndarray) -> np.ndarray:
    """corr[j, i] = V(column i, column j) over the rows observed in both (reference :61-77 builds
    a frame whose COLUMN is the first operand), zero diagonal, absolute value (:21-33)."""
    n, C = codes.shape
    obs = np.asarray(data_m) == 1
    card = codes.max(axis=0) + 1
    out = np.zeros((C, C), dtype=np.float64)
    for i in range(C):
        for j in range(C):
            if i == j:
                continue
            both = obs[:, i] & obs[:, j]
            ci, cj = codes[both, i], codes[both, j]
            table = np.bincount(ci * card[j] + cj, minlength=card[i] * card[j]).reshape(card[i], card[j])
            table = table[table.sum(axis=1) > 0][:, table.sum(axis=0) > 0]      # crosstab keeps seen labels only
            out[j, i] = cramers_v_table(table)
    return np.abs(out)


def association_map_device(codes: np.ndarray, data_m: np.ndarray, device) -> np.ndarray:
    """``association_map`` with the C^2 contingency tables counted on the GPU (ddg_contingency: one pass per column
    pair over the co-observed rows); the chi-square / Cramer's-V arithmetic on each small table is the same host code,
    so the map is bit-identical to ``association_map``."""
    from .. import ops
    n, C = codes.shape
    obs = np.asarray(data_m) == 1
    card = (codes.max(axis=0) + 1).astype(np.int64).tolist() if n else [1] * C
    tabs = ops.contingency_tables(codes, obs, card, device)
    out = np.zeros((C, C), dtype=np.float64)
    for (i, j), t in tabs.items():
        t = t[t.sum(axis=1) > 0][:, t.sum(axis=0) > 0]                  # crosstab keeps seen labels only
        out[j, i] = cramers_v_table(t)
        out[i, j] = cramers_v_table(np.ascontiguousarray(t.T))
    return np.abs(out)


def get_tongji_attn_map(impute_data, method, isNull):
    """reference: utils/get_attr_attn.py:21-33."""
    corr = impute_data if isNull else impute_data.corr(method=method)
    v = np.array(corr.values if hasattr(corr, "values") else corr, dtype=float)
    np.fill_diagonal(v, 0)
    return np.abs(v)


def get_attr_map_Null_kendall(pathMiss, data_m, categorical_cols, continuous_cols, device=None):
    """reference: utils/get_attr_attn.py:35-80 -> corr_map [C,C] float64.  ``device`` (extension): a GPU device
    counts the contingency tables there (same map, bit for bit); None keeps the host bincount path."""
    Miss_data = pd.read_csv(pathMiss)
    codes = discretise(Miss_data, continuous_cols)
    if device is not None and str(device).startswith("cuda"):
        return association_map_device(codes, data_m, device)
    return association_map(codes, data_m)
