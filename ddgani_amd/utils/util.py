"""Host-side mirror of the reference's ``utils/util.py`` for the imputation path: same function
names, arguments, return values and side effects (file:line of the reference function in every
docstring), re-written vectorised so that the data preparation scales to the synthetic tables of
BASELINE.json, with the N x N row attention handed to the HIP library (``ddg_attn_topk_fill``).

The encoders the reference takes from scikit-learn (``OrdinalEncoder``) and the pandas routines
whose tie-breaking defines results (``value_counts``) are called as the reference calls them.
"""
from __future__ import annotations

import random

import numpy as np
import pandas as pd
import torch

from .field import CategoricalField, NumericalField


# ------------------------------------------------------------------------------------------ metrics
def normalization(data, parameters=None):
    """reference: utils/util.py:17-48.  Column-wise (x - min) / (max + 1e-6), max taken after the
    shift; with ``parameters`` the given min / max are applied."""
    data = np.asarray(data, dtype=float)
    if parameters is None:
        min_val = np.nanmin(data, axis=0)
        shifted = data - min_val
        max_val = np.nanmax(shifted, axis=0)
        norm = shifted / (max_val + 1e-6)
        parameters = {"min_val": min_val, "max_val": max_val}
    else:
        norm = (data - parameters["min_val"]) / (parameters["max_val"] + 1e-6)
    return norm, parameters


def errorLoss(imputed_data, ori_data, M, value_cat, continuous_cols, enc):
    """reference: utils/util.py:83-139 -> (ARMSE, AMAE).  Categoricals are ordinal-encoded with
    ``enc``, every column is min-max(+1e-6) normalised by the ground truth, numeric columns score
    RMSE / MAE over their missing cells, categorical columns the mismatch rate (under both
    metrics); the mean runs over the columns that have at least one missing cell."""
    imp = imputed_data.copy()
    ori = ori_data.copy()
    if len(value_cat) != 0:
        imp[value_cat] = enc.transform(imp[value_cat])
        ori[value_cat] = enc.transform(ori[value_cat])
    imp = np.nan_to_num(imp.values.astype(float))
    ori = np.nan_to_num(ori.values.astype(float))
    ori, par = normalization(ori)
    imp, _ = normalization(imp, par)
    miss = 1.0 - np.asarray(M, dtype=float)
    n_miss = miss.sum(axis=0)
    is_num = np.zeros(ori.shape[1], dtype=bool)
    is_num[list(continuous_cols)] = True
    armse = amae = 0.0
    used = 0
    for i in range(ori.shape[1]):
        if n_miss[i] == 0:
            continue
        d = ori[:, i] - imp[:, i]
        if is_num[i]:
            armse += np.sqrt(np.sum(miss[:, i] * d ** 2) / n_miss[i])
            amae += np.sum(miss[:, i] * np.abs(d)) / n_miss[i]
        else:
            r = np.sum(miss[:, i] * (ori[:, i] != imp[:, i]).astype("int")) / n_miss[i]
            armse += r
            amae += r
        used += 1
    return armse / used, amae / used


# ------------------------------------------------------------------------------------------ encoding
def categorical_to_code(miss_data_x, value_cat, enc):
    """reference: utils/util.py:538-550.  Ordinal-encodes the categorical columns IN PLACE (callers
    pass a copy) and returns (frame, encoder); fits a new OrdinalEncoder when ``enc`` is None."""
    if len(value_cat) == 0:
        return miss_data_x, None
    if enc is None:
        from sklearn.preprocessing import OrdinalEncoder
        enc = OrdinalEncoder()
        enc.fit(miss_data_x[value_cat])
    miss_data_x[value_cat] = enc.transform(miss_data_x[value_cat])
    return pd.DataFrame(miss_data_x), enc


def Data_convert(data, model_name, continuous_cols):
    """reference: utils/util.py:52-78 -> (fields, feed_data).  ``data`` has integer column labels;
    numeric columns become one encoded column, categorical columns a one-hot block; feed_data is a
    float64 frame [N, D] with columns 0..D-1."""
    fields = []
    blocks = []
    cont = set(continuous_cols)
    for i, _ in enumerate(list(data)):
        col = np.asarray(data[i])
        if i in cont:
            f = NumericalField(model=model_name)
            f.get_data(data[i])
            f.learn()
            blocks.append(np.asarray(f.convert(col), dtype=float).reshape(-1, 1))
        else:
            f = CategoricalField("one-hot", noise=None)
            f.get_data(data[i])
            f.learn()
            blocks.append(f.convert(col).astype(float))
        fields.append(f)
    feed = np.concatenate(blocks, axis=1) if blocks else np.zeros((len(data), 0))
    return fields, pd.DataFrame(feed)


def field_widths(fields):
    return [int(f.dim()) for f in fields]


def Data_convert_device(data, model_name, continuous_cols, device):
    """``Data_convert`` (utils/util.py:52-78) with the encode itself on the GPU: the fields are learnt on the host
    (min / max or mean / std of a numeric column, sorted distinct values of a categorical one: one pass each), the
    [N, D] table is then written by ddg_encode_table from the [N, C] column values -- the one-hot blocks are never
    materialised on the host (1M x 176 float64 = 1.4 GB for BASELINE config 3).
    -> (fields, code [N, D] fp32 on ``device``), bit-identical to torch.tensor(Data_convert(...)[1].values, dtype=float)."""
    from .. import ops
    fields, cols, mode, scale, shift = [], [], [], [], []
    cont = set(continuous_cols)
    for i, _ in enumerate(list(data)):
        col = np.asarray(data[i])
        if i in cont:
            f = NumericalField(model=model_name)
            f.get_data(data[i])
            f.learn()
            cols.append(np.asarray(col, dtype=np.float64))
            if model_name == "minmax":
                mode.append(0); scale.append(float(f.K)); shift.append(float(f.Min))
            else:
                mode.append(1); scale.append(float(f.sigma)); shift.append(float(f.mu))
        else:
            f = CategoricalField("one-hot", noise=None)
            f.get_data(data[i])
            f.learn()
            cols.append(f.codes(col.reshape(-1)).astype(np.float64))
            mode.append(0); scale.append(1.0); shift.append(0.0)
        fields.append(f)
    dev = torch.device(device)
    layout = ops.FieldLayout(field_widths(fields), dev)
    vals = torch.from_numpy(np.ascontiguousarray(np.stack(cols, axis=1))).to(dev)
    up = lambda v, dt: torch.tensor(np.asarray(v), dtype=dt).to(dev)
    code = ops.encode_table(vals, layout, up(mode, torch.uint8), up(scale, torch.float64), up(shift, torch.float64))
    return fields, code


def get_M_by_data_m(data_m, filed, device):
    """reference: utils/util.py:369-383: the [N,C] mask with every column repeated over the encoded
    width of its field, as a float tensor on ``device``."""
    M = np.repeat(np.asarray(data_m), field_widths(filed), axis=1)
    return torch.tensor(M, dtype=torch.float).to(device)


def reconvert_data(x_, fields, value_cat, values, miss_data_x, data_m, enc):
    """reference: utils/util.py:228-254.  Decode an encoded table (numeric: inverse scaling;
    categorical: arg-max of the block), then keep the observed cells of ``miss_data_x`` and take
    only the missing ones from the decoded table; categorical columns come back as their original
    labels.  Like the reference it ordinal-encodes ``miss_data_x[value_cat]`` in place."""
    x_np = x_.detach().cpu().numpy() if isinstance(x_, torch.Tensor) else np.asarray(x_)
    cols = []
    a = 0
    for f in fields:
        d = f.dim()
        cols.append(np.asarray(f.reverse(x_np[:, a:a + d])).reshape(-1))
        a += d
    current = np.stack(cols, axis=1) if cols else np.zeros((x_np.shape[0], 0))
    if value_cat:
        miss_data_x[value_cat] = enc.transform(miss_data_x[value_cat])
    m = np.asarray(data_m)
    merged = pd.DataFrame(miss_data_x.values * m + current * (1 - m))
    if value_cat:
        merged.columns = values
        merged[value_cat] = enc.inverse_transform(merged[value_cat])
    return merged


def test_impute_data_rmse(x_code, fields, value_cat, values, miss_data_x, data_m, enc, ori_data, continuous_cols):
    """reference: utils/util.py:648-658 -> (ARMSE, AMAE) of an encoded table."""
    impute_data = pd.DataFrame(reconvert_data(x_code, fields, value_cat, values, miss_data_x, data_m, enc))
    impute_data.columns = values
    return errorLoss(impute_data, ori_data, data_m, value_cat, continuous_cols, enc)


class DeviceScorer:
    """``test_impute_data_rmse`` (utils/util.py:648-658 = reconvert_data + errorLoss) for encoded tables that live on
    the GPU: everything that does not depend on the table being scored -- the normalised ground truth, the decode
    constants of the fields, the missing-cell counts -- is prepared once; ``score(code)`` is then one pass over the
    missing cells on the device (ddg_score_table) and a C-element finish on the host, instead of a pandas round trip of
    the whole table per sampling pass (model/Diffusion.py:617)."""

    def __init__(self, fields, value_cat, values, ori_data, data_m, enc, continuous_cols, device):
        from .. import ops
        ori = ori_data.copy()
        if len(value_cat) != 0:
            ori[value_cat] = enc.transform(ori[value_cat])
        ori = np.nan_to_num(ori.values.astype(float))
        ori_n, par = normalization(ori)
        widths = field_widths(fields)
        C = len(fields)
        self.layout = ops.FieldLayout(widths, device)
        is_num = np.zeros(C, dtype=bool)
        is_num[list(continuous_cols)] = True
        if not np.array_equal(is_num, np.asarray(widths) == 1):
            # a categorical column with a single distinct value (or a numeric one encoded wider) cannot be told apart
            # by width on the device
            raise ValueError("DeviceScorer: numeric columns must be exactly the width-1 fields")
        K = np.ones(C)
        mn = np.zeros(C)
        f32 = np.zeros(C, dtype=np.uint8)
        pos = np.zeros(sum(widths))
        a = 0
        for c, f in enumerate(fields):
            if widths[c] == 1:
                K[c], mn[c] = float(f.K), float(f.Min)
                # NumericalField.reverse is ``features / K + Min`` on the fp32 code: numpy keeps fp32 when K and Min are
                # Python floats (learnt from an object-typed column) and promotes to fp64 for numpy float64 scalars
                weak = [not isinstance(v, np.generic) for v in (f.K, f.Min)]
                if weak[0] != weak[1]:
                    raise ValueError("DeviceScorer: K and Min of a field have different scalar kinds")
                f32[c] = 1 if weak[0] else 0
            else:
                pos[a:a + widths[c]] = np.asarray(f._values, dtype=float)
            a += widths[c]
        up = lambda v: torch.tensor(np.ascontiguousarray(v), dtype=torch.float64).to(device)
        m = np.asarray(data_m)
        self.m8 = torch.tensor(m != 0, dtype=torch.uint8).to(device)
        self.ori_n, self.K, self.mn, self.pos = up(ori_n), up(K), up(mn), up(pos)
        self.f32 = torch.tensor(f32).to(device)
        self.nmin, self.nden = up(par["min_val"]), up(par["max_val"] + 1e-6)
        self.n_miss = (1.0 - m.astype(float)).sum(axis=0)
        self.is_num = is_num

    def decode(self, code: torch.Tensor, miss_data_x, value_cat, values, enc):
        """``reconvert_data`` (utils/util.py:228-254) of an encoded table on the device: one pass of ddg_decode_table
        (observed cells of ``miss_data_x``, decoded cells elsewhere), then the categorical codes back to labels on
        the host.  Like the reference it ordinal-encodes ``miss_data_x[value_cat]`` in place.  -> DataFrame [N, C]"""
        from .. import ops
        if value_cat:
            miss_data_x[value_cat] = enc.transform(miss_data_x[value_cat])
        obs = torch.from_numpy(np.ascontiguousarray(miss_data_x.values.astype(np.float64))).to(code.device)
        out = ops.decode_table(code.contiguous(), self.m8, obs, self.layout, self.K, self.mn, self.f32, self.pos)
        merged = pd.DataFrame(out.cpu().numpy())
        if value_cat:
            merged.columns = values
            merged[value_cat] = enc.inverse_transform(merged[value_cat])
        return merged

    def score(self, code: torch.Tensor):
        """-> (ARMSE, AMAE) of the encoded table ``code`` [N,D] (fp32, on the scorer's device)."""
        from .. import ops
        s = ops.score_table(code.contiguous(), self.ori_n, self.m8, self.layout, self.K, self.mn, self.f32, self.pos, self.nmin,
                            self.nden).cpu().numpy()
        armse = amae = 0.0
        used = 0
        for i in range(len(self.n_miss)):
            if self.n_miss[i] == 0:
                continue
            if self.is_num[i]:
                armse += np.sqrt(s[i, 0] / self.n_miss[i])
                amae += s[i, 1] / self.n_miss[i]
            else:
                armse += s[i, 2] / self.n_miss[i]
                amae += s[i, 2] / self.n_miss[i]
            used += 1
        return armse / used, amae / used


def sort_corr(corr_map):
    """reference: utils/util.py:692-697: per column, the other columns by descending weight
    (stable, like ``sorted(..., reverse=True)``)."""
    out = {}
    for index, row in enumerate(np.asarray(corr_map)):
        out[index] = sorted(range(len(row)), key=lambda x: row[x], reverse=True)
    return out


# ------------------------------------------------------------------------------------------ attention fill
def attention_inputs(corr_map, miss_data, data_m, categorical_cols, enc, value_cat, device=None):
    """Host preparation of init_attn_2 (utils/util.py:390-448): mode-fill 'Null' categoricals,
    ordinal-encode, z-score / one-hot encode ("mean_std"), expand mask and column weights.
    Returns (fields, X [N,D] fp32, M [N,D] fp32, W [C,D] fp32, offsets).  With a GPU ``device`` the encode runs there
    (ddg_encode_table) and X, M, W are device tensors."""
    code = miss_data.copy()
    names = list(miss_data.columns)
    con_cols = [i for i in range(len(names)) if i not in categorical_cols]
    for i in categorical_cols:
        col = names[i]
        for val in miss_data[col].value_counts().index.tolist():
            if val != "Null":
                code[col] = code[col].where(code[col] != "Null", val)
                break
    code, enc = categorical_to_code(code, value_cat, enc)
    code.columns = list(range(code.shape[1]))
    if device is not None and torch.device(device).type == "cuda":
        dev = torch.device(device)
        fields, X = Data_convert_device(code, "mean_std", con_cols, dev)
        widths = field_widths(fields)
        wt = torch.tensor(widths, device=dev)
        M = torch.repeat_interleave(torch.tensor(np.asarray(data_m), dtype=torch.float).to(dev), wt, dim=1)
        W = torch.repeat_interleave(torch.tensor(np.asarray(corr_map), dtype=torch.float).to(dev), wt, dim=1)
        return fields, X, M, W, widths
    fields, enc_code = Data_convert(code, "mean_std", con_cols)
    widths = field_widths(fields)
    X = torch.tensor(enc_code.values, dtype=torch.float)
    M = torch.tensor(np.repeat(np.asarray(data_m), widths, axis=1), dtype=torch.float)
    W = torch.tensor(np.repeat(np.asarray(corr_map), widths, axis=1), dtype=torch.float)
    return fields, X, M, W, widths


def topk_count(n_rows: int, top_k) -> int:
    """k of the reference (utils/util.py:459): ``int(N * top_k)``.  Extension needed by BASELINE.json config 3
    ("top_k=10"): an INTEGER top_k >= 1 (``"top_k": 10`` in param.json) is an absolute count; any float, 1.0
    included, is the reference's fraction."""
    if isinstance(top_k, (int, np.integer)) and not isinstance(top_k, bool) and top_k >= 1:
        return int(top_k)
    return int(n_rows * top_k)


def init_attn_2(corr_map, miss_data, data_m, categorical_cols, enc, value_cat, device, top_k):
    """reference: utils/util.py:386-501 -> (DataFrame [N,C] in the original domain, code tensor
    [N,D] on ``device``).  The C dense N x N softmax / top-k matrices of the reference
    (:449-466, :484-496) are replaced by one call of the streaming HIP kernel."""
    if corr_map is None:
        # The reference's unweighted branch (utils/util.py:467-476) takes ``topk(...).values[:, -1]`` (wrong axis: a
        # [k] vector of the LAST query's k largest probabilities) and expands it to [N, N]: torch raises
        # "The expanded size of the tensor (N) must match the existing size (k)" for every 1 < k < N, i.e. for every
        # top_k the reference is configured with.  Same error behaviour here.  Only k == 1 and k == N pass that line in
        # the reference; their result is an artefact of the wrong axis and is not on the DDGANI path
        # (BaseLine/DiffGANI.py:21 always passes a map).
        n = len(miss_data)
        k = int(n * top_k)
        if 1 < k < n:
            raise RuntimeError(f"The expanded size of the tensor ({n}) must match the existing size ({k}) at non-singleton "
                               f"dimension 0.  Target sizes: [{n}, {n}].  Tensor sizes: [{k}, 1]  (init_attn_2 with "
                               "corr_map=None fails here in the reference too: utils/util.py:472-473)")
        raise NotImplementedError("init_attn_2(corr_map=None) with int(N*top_k) in {0, 1, N}: degenerate cases of the "
                                  "reference's unweighted branch (utils/util.py:467-476) are not built")
    from .. import ops
    values = miss_data.columns
    dev = torch.device(device)
    fields, X, M, W, widths = attention_inputs(corr_map, miss_data, data_m, categorical_cols, enc, value_cat, device=dev)
    k = topk_count(X.shape[0], top_k)
    if k < 1:
        raise IndexError("init_attn_2: int(N * top_k) == 0 (the reference fails at values[-1] here)")
    lay = ops.FieldLayout(widths, dev)
    X0 = (X * M).to(dev)
    m8 = torch.tensor(np.asarray(data_m) != 0, dtype=torch.uint8).to(dev)
    new_code = ops.attn_topk_fill(X0, m8, X0, m8, W.to(dev), lay, k)
    impute_data = pd.DataFrame(reconvert_data(new_code, fields, value_cat, values, miss_data, data_m, enc))
    impute_data.columns = values
    return impute_data, new_code


# ------------------------------------------------------------------------------------------ training helpers
def sample_x(x, batch_size):
    """reference: utils/util.py:582-585: ``batch_size`` distinct rows drawn with numpy's global RNG."""
    rows = list(np.random.choice(x.size()[0], batch_size, replace=False))
    return x[rows, :], rows


def get_valid_data_index(data_m, discriminator, impute_data_code, device, rng="host", seed=0):
    """reference: utils/util.py:623-646.  20 % of the rows for the Learner's validation split:
    sampled (python ``random``) from the complete rows when there are enough of them, otherwise the
    complete rows plus the rows a freshly trained plain discriminator ``D`` scores lowest.
    On a GPU ``device`` the 1 000 full-batch Adam steps of that discriminator run in the HIP library
    (ddg_dplain_grads / ddg_adam_step / ddg_dplain_fwd).  rng="host": its dropout keeps are drawn from torch's CPU
    generator exactly as the reference's CPU run draws them (two ``bernoulli_(0.8)`` of [N, input_dim] per pass, 1 001
    passes), so a run seeded like the reference follows the reference's random stream; rng="chip": Philox on the
    device (``seed``).  A CPU ``device`` keeps stock torch (the oracle-side path of the CPU tests)."""
    data_m = np.asarray(data_m)
    n = data_m.shape[0]
    valid = np.nonzero(np.all(data_m == 1, axis=1))[0].tolist()
    want = int(0.2 * n)
    if len(valid) >= want:
        return random.sample(valid, want)
    from ..model.Discriminator_model import D
    cur = D(discriminator.input_dim, discriminator.latent_dim, discriminator.out_dim)      # (CPU: the reference's init draws)
    dev = torch.device(device)
    if dev.type != "cuda":
        from torch import optim
        cur = cur.to(dev)
        opt = optim.Adam(cur.parameters(), lr=0.002)
        cur.train()
        m = torch.tensor(data_m).float().to(dev)
        x = impute_data_code.to(dev)
        for _ in range(1000):
            opt.zero_grad()
            p = cur(x)
            loss = -torch.mean(m * torch.log(p + 1e-8) + (1 - m) * torch.log(1 - p + 1e-8))
            loss.backward()
            opt.step()
        p = cur(x)
    else:
        from .. import ops
        P = ops.DplainParams(cur.state_dict(), dev)
        buf = ops.ParamGrads(P, ops.DPLAIN_KEYS, n_extra=0)
        m = torch.tensor(data_m).float().to(dev).contiguous()
        x = impute_data_code.to(dev).float().contiguous()
        width = P.in_dim

        def keeps():
            if rng != "host":
                return None
            return [torch.empty(n, width).bernoulli_(0.8).to(torch.uint8).to(dev) for _ in range(2)]
        for step in range(1000):
            ops.dplain_grads(P, buf, x, m, drops=keeps(), seed=seed, stream_id=step + 1)
            buf.adam_step(lr=0.002, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
        p = ops.dplain_fwd(P, x, drops=keeps(), seed=seed, stream_id=1001)
    _, order = torch.sort(p.sum(dim=1))
    return np.union1d(valid, order[:want].cpu().numpy())


# ------------------------------------------------------------------------------------------ missingness
def MAR(data, continuous_cols, miss_seed):
    """reference: utils/util.py:701-729.  Rows sorted by one random numeric column; a random
    contiguous run of 40 % of them loses each other cell with probability 0.5."""
    np.random.seed(miss_seed)
    n, dim = data.shape
    pick = np.random.choice(continuous_cols)
    order = np.argsort(data.iloc[:, pick].values)
    count = int(n * 0.4)
    start = np.random.randint(0, n - count + 1)
    drop = np.random.choice([True, False], size=(count, dim), p=[0.5, 0.5])
    drop[:, pick] = False
    data_m = (~pd.isna(data).values).astype(float)
    rows = order[start:start + count]
    sub = data_m[rows]
    sub[drop] = 0
    data_m[rows] = sub
    return data_m


def MNAR(data, continuous_cols, categorical_cols, miss_seed):
    """reference: utils/util.py:732-762.  Categorical columns lose cells with probability 0.2;
    numeric columns lose 40 % of the cells at or below the median."""
    np.random.seed(miss_seed)
    n, dim = data.shape
    data_m = np.ones((n, dim))
    for i in range(dim):
        if i in categorical_cols:
            mask = np.random.choice([True, False], size=n, p=[0.2, 0.8])
            data_m[mask, i] = 0
        else:
            col = data.iloc[:, i].values
            low = np.where(col <= np.median(col))[0]
            chosen = np.random.choice(low, int(len(low) * 0.4), replace=False)
            data_m[chosen, i] = 0
    return data_m


def Region(data, miss_seed):
    """reference: utils/util.py:766-786.  One rectangle holding 20 % of the cells."""
    np.random.seed(miss_seed)
    n, dim = data.shape
    missing = int(n * dim * 0.2)
    rows, cols = missing, 1
    while rows > n:
        rows //= 2
        cols = missing // rows
    r0 = np.random.randint(0, n - rows + 1)
    c0 = np.random.randint(0, dim - cols + 1)
    data_m = np.ones((n, dim))
    data_m[r0:r0 + rows, c0:c0 + cols] = 0
    return data_m
