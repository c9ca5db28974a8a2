"""GPU: SURVEY.md 8(f) rows 2 and 3 through the C-ABI.
 * ddg_contingency (the C^2 crosstabs of get_attr_map_Null_kendall, utils/get_attr_attn.py:61-77): integer tables equal
   to the host bincount tables, the resulting column-association map bit-identical to the host mirror and within 1e-12
   of the REFERENCE's map on `hospital` (tests/golden/host_hospital.npz);
 * ddg_score_table (reconvert_data + errorLoss, utils/util.py:228-254, :83-139): ARMSE / AMAE of an encoded table
   equal to the host mirror's and to the reference's values (golden decode fixture; the reference's own sampling
   passes of the end-to-end run) to 1e-12."""
import json
import os
import random

import numpy as np
import pandas as pd
import pytest
import torch

from util_golden import GOLD, load

pytestmark = pytest.mark.gpu
CAT = [0, 1, 2, 3, 4, 5, 6]


@pytest.fixture(scope="module")
def prepared(tmp_path_factory):
    from ddgani_amd.utils import util
    from ddgani_amd.utils.data_loader import data_loader, set_label, value_loader
    work = str(tmp_path_factory.mktemp("tab")) + "/"
    label_file, label_num = set_label(os.path.join(GOLD, "hospital.csv"), work, "True")
    np.random.seed(0); random.seed(0)
    r = data_loader(label_file, 0.2, CAT, "True", "MCAR", 42)
    values, _, value_cat = value_loader(label_file, r[6], "True")
    r[1].columns = values
    r[1].to_csv(work + "miss.csv", index=None)
    _, enc = util.categorical_to_code(r[9], value_cat, enc=None)
    return dict(work=work, r=r, values=values, value_cat=value_cat, enc=enc)


def test_association_map_on_device_matches_reference(prepared):
    from ddgani_amd.utils import get_attr_attn as ga
    dev = torch.device("cuda:0")
    z = load("host_hospital")
    r = prepared["r"]
    host = ga.get_attr_map_Null_kendall(prepared["work"] + "miss.csv", r[2], CAT, r[6])
    devm = ga.get_attr_map_Null_kendall(prepared["work"] + "miss.csv", r[2], CAT, r[6], device=dev)
    assert np.array_equal(host, devm)                                   # same tables -> same map, bit for bit
    np.testing.assert_allclose(devm, z["corr"], rtol=1e-12, atol=0)     # the reference's own map


@pytest.mark.parametrize("N,cards", [(5000, [20, 20, 3, 7, 20]), (200_000, [20, 150, 120, 2, 20, 20, 64]),
                                     (70_001, [1, 20, 300])])
def test_contingency_tables_match_bincount(N, cards):
    """synthetic codes incl. tables larger than the LDS budget (150 x 120 cells -> global atomics), several row
    chunks per pair, a single-valued column and ragged sizes"""
    from ddgani_amd import ops
    rng = np.random.default_rng(N)
    C = len(cards)
    codes = np.stack([rng.integers(0, c, N) for c in cards], axis=1)
    obs = rng.uniform(size=(N, C)) < 0.8
    tabs = ops.contingency_tables(codes, obs, cards, torch.device("cuda:0"))
    assert len(tabs) == C * (C - 1) // 2
    for (i, j), t in tabs.items():
        both = obs[:, i] & obs[:, j]
        ref = np.bincount(codes[both, i] * cards[j] + codes[both, j], minlength=cards[i] * cards[j])
        assert np.array_equal(t, ref.reshape(cards[i], cards[j])), (i, j)


def test_device_score_matches_host_and_reference(prepared):
    from ddgani_amd.BaseLine import Mean
    from ddgani_amd.utils import util
    dev = torch.device("cuda:0")
    z = load("host_hospital")
    r, values, value_cat, enc = prepared["r"], prepared["values"], prepared["value_cat"], prepared["enc"]
    data_m, con = r[2], r[6]
    filled = Mean.fill_data_mean(r[7].copy(), con)
    fdf = pd.DataFrame(filled, columns=values)
    code, _ = util.categorical_to_code(fdf.copy(), value_cat, enc)
    code.columns = list(range(code.shape[1]))
    fields, _ = util.Data_convert(code.copy(), "minmax", con)
    scorer = util.DeviceScorer(fields, value_cat, values, r[0], data_m, enc, con, dev)
    x = torch.tensor(z["decode_in"]).to(dev)
    armse, amae = scorer.score(x)
    assert abs(armse - float(z["decode_armse"])) < 1e-12 and abs(amae - float(z["decode_amae"])) < 1e-12
    # random encoded tables (ties in a block: first maximum wins, as np.argmax) against the host mirror
    g = torch.Generator().manual_seed(3)
    for _ in range(3):
        xr = torch.rand(x.shape, generator=g)
        xr[::7] = torch.round(xr[::7] * 4) / 4                          # exact ties inside one-hot blocks
        h = util.test_impute_data_rmse(xr, fields, value_cat, values, fdf.copy(), data_m, enc, r[0], con)
        d = scorer.score(xr.to(dev))
        assert abs(h[0] - d[0]) < 1e-12 and abs(h[1] - d[1]) < 1e-12, (h, d)


def test_device_score_on_reference_sampling_passes(tmp_path):
    """the merged code of the reference's own sampling passes (epochs 100 / 200 of the hospital run) scores to the
    reference's ARMSE / AMAE"""
    from ddgani_amd.utils import util
    from ddgani_amd.utils.data_loader import data_loader, set_label, value_loader
    dev = torch.device("cuda:0")
    z, zp = load("e2e_hospital"), load("e2e_hospital_passes")
    param = json.loads(str(z["param_json"]))
    path = str(tmp_path) + "/"
    label_file, _ = set_label(os.path.join(GOLD, "hospital.csv"), path, "True")
    np.random.seed(0); random.seed(0); torch.manual_seed(0)
    (ori_data, _, data_m, _, _, _, con, _, _, all_data) = data_loader(label_file, float(z["miss_rate"]),
                                                                      param["categorical_cols"], "True", "MCAR",
                                                                      int(z["seed"]))
    values, _, value_cat = value_loader(label_file, con, "True")
    _, enc = util.categorical_to_code(all_data, value_cat, enc=None)
    attn = pd.DataFrame(z["attn_coded"], columns=values)
    attn[value_cat] = enc.inverse_transform(attn[value_cat])
    coded, _ = util.categorical_to_code(attn.copy(), value_cat, enc)
    coded.columns = list(range(coded.shape[1]))
    fields, _ = util.Data_convert(coded, param["model_name"], con)
    scorer = util.DeviceScorer(fields, value_cat, values, ori_data, data_m, enc, con, dev)
    for i in range(int(zp["n_passes"])):
        armse, amae = scorer.score(torch.tensor(zp[f"p{i}__code"]).to(dev))
        rr, mm = zp[f"p{i}__scores"]
        assert abs(armse - rr) < 1e-12 and abs(amae - mm) < 1e-12, (armse, rr, amae, mm)


def test_device_encode_and_decode_equal_the_host_mirror(prepared):
    """(f3) ddg_encode_table == torch.tensor(Data_convert(...).values, dtype=float) bit for bit, in both encodings of the
    path ("minmax": BaseLine/DiffGANI.py:27-29, "mean_std": utils/util.py:407), and against the reference's own encode
    arrays (tests/golden/host_hospital.npz); ddg_decode_table + label mapping == reconvert_data (utils/util.py:228-254)
    cell for cell, on encoded tables with exact ties inside one-hot blocks."""
    from ddgani_amd.BaseLine import Mean
    from ddgani_amd.utils import util
    dev = torch.device("cuda:0")
    z = load("host_hospital")
    r, values, value_cat, enc = prepared["r"], prepared["values"], prepared["value_cat"], prepared["enc"]
    data_m, con = r[2], r[6]
    filled = Mean.fill_data_mean(r[7].copy(), con)
    fdf = pd.DataFrame(filled, columns=values)
    code, _ = util.categorical_to_code(fdf.copy(), value_cat, enc)
    code.columns = list(range(code.shape[1]))
    for model in ("minmax", "mean_std"):
        fields_h, feed = util.Data_convert(code.copy(), model, con)
        host = torch.tensor(feed.values, dtype=torch.float)
        fields_d, devc = util.Data_convert_device(code.copy(), model, con, dev)
        assert devc.dtype == torch.float32 and tuple(devc.shape) == tuple(host.shape)
        assert torch.equal(devc.cpu(), host), float((devc.cpu() - host).abs().max())
        assert [f.dim() for f in fields_d] == [f.dim() for f in fields_h]
    for key in ("encode_minmax", "feed_minmax", "x_minmax"):          # the reference's own array, when the fixture holds it
        if key in z.files:
            _, devc = util.Data_convert_device(code.copy(), "minmax", con, dev)
            assert np.array_equal(devc.cpu().numpy(), z[key].astype(np.float32))
            break
    fields, _ = util.Data_convert(code.copy(), "minmax", con)
    scorer = util.DeviceScorer(fields, value_cat, values, r[0], data_m, enc, con, dev)
    g = torch.Generator().manual_seed(5)
    x = torch.tensor(z["decode_in"])
    for i in range(3):
        xr = x if i == 0 else torch.rand(x.shape, generator=g)
        if i == 2:
            xr[::5] = torch.round(xr[::5] * 4) / 4                      # exact ties: first maximum wins, as np.argmax
        host = util.reconvert_data(xr, fields, value_cat, values, fdf.copy(), data_m, enc)
        got = scorer.decode(xr.to(dev), fdf.copy(), value_cat, values, enc)
        assert list(got.columns) == list(host.columns)
        for c in host.columns:
            if c in value_cat:
                assert (got[c].values == host[c].values).all(), c
            else:
                assert np.array_equal(got[c].values.astype(float), host[c].values.astype(float)), c
