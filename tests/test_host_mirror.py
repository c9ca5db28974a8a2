"""CPU: the host-side mirror (ddgani_amd/utils, BaseLine/Mean) against golden outputs of the
reference's own functions on its ``hospital`` table (tests/golden/host_hospital.npz, produced by
tests/golden/make_host_golden.py by calling the reference).  Integer / mask / label results are
bit-exact; float results within 1e-12 (same formulas in float64)."""
import os
import random

import numpy as np
import pandas as pd
import pytest
import torch

from util_golden import GOLD, load

CAT = [0, 1, 2, 3, 4, 5, 6]


@pytest.fixture(scope="module")
def prepared(tmp_path_factory):
    from ddgani_amd.utils import util
    from ddgani_amd.utils.data_loader import data_loader, set_label, value_loader
    work = str(tmp_path_factory.mktemp("host")) + "/"
    label_file, label_num = set_label(os.path.join(GOLD, "hospital.csv"), work, "True")
    np.random.seed(0); random.seed(0)
    r = data_loader(label_file, 0.2, CAT, "True", "MCAR", 42)
    values, value_num, value_cat = value_loader(label_file, r[6], "True")
    r[1].columns = values
    r[1].to_csv(work + "miss.csv", index=None)
    _, enc = util.categorical_to_code(r[9], value_cat, enc=None)
    return dict(work=work, label_file=label_file, label_num=label_num, r=r, values=values, value_cat=value_cat, enc=enc)


@pytest.mark.parametrize("mt", ["MCAR", "MAR", "MNAR", "Region"])
def test_missingness_masks(prepared, mt):
    from ddgani_amd.utils.data_loader import data_loader
    z = load("host_hospital")
    np.random.seed(0); random.seed(0)
    r = data_loader(prepared["label_file"], 0.3, CAT, "True", mt, 7)
    assert np.array_equal(np.asarray(r[2]).astype(np.int8), z[f"data_m_{mt}"])
    assert list(r[0].iloc[0].astype(str).values) == list(z[f"row0_{mt}"])      # same split + shuffle
    assert prepared["label_num"] == int(z["label_num"])


def test_column_association_map(prepared):
    from ddgani_amd.utils.get_attr_attn import cramers_v, get_attr_map_Null_kendall
    z = load("host_hospital")
    r = prepared["r"]
    corr = get_attr_map_Null_kendall(prepared["work"] + "miss.csv", r[2], CAT, r[6])
    np.testing.assert_allclose(corr, z["corr"], rtol=1e-12, atol=0)
    assert np.all(np.diag(corr) == 0) and corr.max() < 0.05            # the +1e5 damping keeps weights in the 1e-3..2e-2 range
    # 2x2 tables take Yates' correction; degenerate tables give 0
    assert cramers_v([0, 0, 1, 1, 0, 1], [0, 0, 1, 1, 1, 0]) >= 0
    assert cramers_v([1, 1, 1], [0, 1, 2]) == 0


def test_mean_fill_encode_decode_score(prepared):
    from ddgani_amd.BaseLine import Mean
    from ddgani_amd.utils import util
    z = load("host_hospital")
    r, values, value_cat, enc = prepared["r"], prepared["values"], prepared["value_cat"], prepared["enc"]
    data_m, con = r[2], r[6]
    assert np.array_equal(np.asarray(data_m).astype(np.int8), z["data_m"])
    filled = Mean.fill_data_mean(r[7].copy(), con)
    fdf = pd.DataFrame(filled, columns=values)
    code, _ = util.categorical_to_code(fdf.copy(), value_cat, enc)
    np.testing.assert_allclose(code.values.astype(float), z["meanfill_coded"], rtol=1e-12)
    code.columns = list(range(code.shape[1]))
    for model in ("minmax", "mean_std"):
        fields, feed = util.Data_convert(code.copy(), model, con)
        assert [f.dim() for f in fields] == z[f"widths_{model}"].tolist()
        np.testing.assert_allclose(feed.values, z[f"feed_{model}"], rtol=1e-12, atol=1e-15)
    fields, feed = util.Data_convert(code.copy(), "minmax", con)
    M = util.get_M_by_data_m(data_m, fields, torch.device("cpu")).numpy()
    assert np.array_equal(M, z["M_expanded"])
    dec = pd.DataFrame(util.reconvert_data(torch.tensor(z["decode_in"]), fields, value_cat, values, fdf.copy(), data_m, enc))
    dec.columns = values
    armse, amae = util.errorLoss(dec, r[0], data_m, value_cat, con, enc)
    assert abs(armse - float(z["decode_armse"])) < 1e-12 and abs(amae - float(z["decode_amae"])) < 1e-12
    dc = dec.copy()
    dc[value_cat] = enc.transform(dc[value_cat])
    np.testing.assert_allclose(dc.values.astype(float), z["decode_coded"], rtol=1e-6)


def test_error_loss_golden():
    """errorLoss itself against the reference's value on a synthetic table (tests/golden/error_loss.npz)."""
    from ddgani_amd.utils import util
    z = load("error_loss")
    imp, ori = pd.DataFrame(z["imputed"]), pd.DataFrame(z["ori"])
    armse, amae = util.errorLoss(imp, ori, z["data_m"], [], z["continuous_cols"].tolist(), None)
    assert abs(armse - float(z["armse"])) < 1e-12 and abs(amae - float(z["amae"])) < 1e-12


def test_attention_inputs_match_reference_encoding():
    """attention_inputs() reproduces the X / W the reference's init_attn_2 builds (golden attn fixtures)."""
    from ddgani_amd.utils import util
    from sklearn.preprocessing import OrdinalEncoder
    for name in ("attn_mixed257", "attn_onehot200", "attn_num64"):
        z = load(name)
        C = z["data_m"].shape[1]
        cat_cols = z["categorical_cols"].tolist()
        cols = {}
        for c in range(C):
            cols[f"a{c}"] = z["cat_values"][:, c].astype(object) if c in cat_cols else z["num_values"][:, c]
        df = pd.DataFrame(cols)
        value_cat = [f"a{c}" for c in cat_cols]
        enc = OrdinalEncoder().fit(df[value_cat]) if cat_cols else None
        fields, X, M, W, widths = util.attention_inputs(z["corr"], df, z["data_m"], cat_cols, enc, value_cat)
        assert np.cumsum([0] + widths).tolist() == z["offsets"].tolist()
        assert np.array_equal(X.numpy(), z["X"]) and np.array_equal(W.numpy(), z["W"])
        assert util.topk_count(X.shape[0], float(z["top_k"])) == int(z["k"])
    assert util.topk_count(1000, 10) == 10 and util.topk_count(1000, 0.05) == 50


def test_field_protocol_and_errors():
    from ddgani_amd.utils.field import CategoricalField, NumericalField
    f = NumericalField("minmax"); f.get_data(np.array([2.0, 4.0, 10.0])); f.learn()
    assert np.allclose(f.reverse(f.convert(np.array([3.0, 10.0]))), [3.0, 10.0]) and f.dim() == 1
    c = CategoricalField("one-hot", noise=None); c.get_data(np.array([2.0, 0.0, 2.0, 5.0])); c.learn()
    assert c.dim() == 3 and c.rev_dict == {0: 0.0, 1: 2.0, 2: 5.0}
    oh = c.convert(np.array([5.0, 0.0]))
    assert oh.tolist() == [[0, 0, 1], [1, 0, 0]] and c.reverse(oh.astype(float)).ravel().tolist() == [5.0, 0.0]
    with pytest.raises(KeyError):
        c.convert(np.array([7.0]))
    with pytest.raises(ValueError):
        NumericalField("gmm")


def test_init_attn_2_without_map_fails_like_the_reference():
    """init_attn_2(corr_map=None): the reference's unweighted branch raises RuntimeError at its expand_as for every
    1 < int(N*top_k) < N (probed on the reference: top_k 0.05 and 0.2 on a 60-row table); so does the mirror."""
    from ddgani_amd.utils import util
    rng = np.random.default_rng(0)
    df = pd.DataFrame({"a": rng.normal(size=60), "b": rng.normal(size=60), "c": rng.normal(size=60)})
    dm = pd.DataFrame((rng.uniform(size=(60, 3)) < 0.8).astype(int))
    for tk, k in ((0.05, 3), (0.2, 12)):
        with pytest.raises(RuntimeError, match=rf"must match the existing size \({k}\)"):
            util.init_attn_2(None, df.copy(), dm, [], None, [], torch.device("cpu"), tk)
    with pytest.raises(NotImplementedError):
        util.init_attn_2(None, df.copy(), dm, [], None, [], torch.device("cpu"), 1.0)
