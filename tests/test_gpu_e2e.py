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
 * ARMSE / AMAE after 100 and 200 epochs inside the distribution of the reference's OWN scores on this run when only
   the fp32 summation order of its CPU GEMMs changes: tests/golden/e2e_hospital_spread.json (made by
   tests/golden/make_e2e_spread.py) holds the reference's scores with 1 to 8 intra-op threads: ARMSE after 100 epochs
   0.34966 ... 0.35229 (spread 2.6e-3, standard deviation 8.7e-4), after 200 epochs 0.26182 ... 0.26523 (spread 3.4e-3,
   standard deviation 1.0e-3); its weights drift apart by 2e-5 after 2 epochs and 1e-3 after 100.  1 200 / 2 400 Adam
   steps through dropout and ReLU amplify any last-bit difference, so "identical to 3 decimals" is not a property the
   reference has against itself on this run.  Asserted: every score within 3 standard deviations of the mean of the
   reference's runs.  The 3-decimal bar is asserted where it CAN hold: on the imputation path in isolation
   (test_imputation_path_on_reference_weights below).
"""
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


def _run_pipeline(tmp_path, tag, csv_name, rng="host", epochs=None, attn_from_reference=False):
    from ddgani_amd.BaseLine import DiffGANI
    from ddgani_amd.utils import util
    from ddgani_amd.utils.data_loader import data_loader, set_label, value_loader
    z = load(tag)
    param = json.loads(str(z["param_json"]))
    param["epochs"] = int(epochs or z["epochs"])
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
        if epoch == 1 and "valid_index" in z.files:
            # get_valid_data_index: hospital has fewer than 20 % complete rows, so the fallback discriminator is trained
            # (1 000 full-batch Adam steps in the HIP library, dropout keeps from the reference's CPU stream) and the 160
            # rows it scores lowest join the split (utils/util.py:623-646).  The selection is not reproducible to the row
            # even by torch itself: the same CPU op sequence with 1 / 2 / 8 intra-op threads selects sets that differ in
            # 11 - 13 entries (tools/valid_index_sensitivity.py; rows near the cut-off swap).  Bound: 40 entries
            # (20 of 270 rows); measured 28.
            pins["valid_rows_differing"] = int(len(np.setxor1d(np.asarray(vidx, dtype=np.int64), z["valid_index"])))
        if epoch in (1, 2, 10):
            for nm, mine in ((f"G_b4_ep{epoch}", G.linears[12].bias), (f"G_temb_ep{epoch}", G.time_embedding.weight),
                             (f"D_b6_ep{epoch}", Dn.linears[6].bias)):
                pins[nm] = float(np.abs(mine.detach().cpu().numpy() - z[nm]).max())
    own_attn = []
    orig_attn = DiffGANI.init_attn_2
    if attn_from_reference:
        # the attention stage runs and is compared as usual, but the diffusion stage is fed the table the REFERENCE's
        # attention stage produced: where near-tie cells legitimately differ (adult), the training-trajectory pins then
        # test the training arithmetic on identical inputs
        def patched(*a, **k):
            df, code = orig_attn(*a, **k)
            own_attn.append(df.copy())
            ref = pd.DataFrame(z["attn_coded"], columns=values)
            if value_cat:
                ref[value_cat] = enc.inverse_transform(ref[value_cat])
            for c in values:
                if c not in value_cat:
                    ref[c] = ref[c].astype(df[c].dtype)
            return ref, code
        DiffGANI.init_attn_2 = patched
    try:
        fill = DiffGANI.get_Diff_acc_RMSE(nan_data, path, miss_rate, miss_data, enc, data_m, param["categorical_cols"],
                                          continuous_cols, value_cat, torch.device("cuda:0"), param, label_data, values,
                                          ori_data, label_num, args, rng=rng, log=log,
                                          epoch_hook=hook if rng == "host" else None)
    finally:
        DiffGANI.init_attn_2 = orig_attn
    fill_df = pd.DataFrame(fill, columns=values)
    final = util.errorLoss(fill_df, ori_data, data_m, value_cat, continuous_cols, enc)
    attn = own_attn[0] if own_attn else pd.read_csv(path + f"Attention_Input_miss_data_{miss_rate}.csv")
    if value_cat:
        attn[value_cat] = enc.transform(attn[value_cat])
    return z, data_m, attn.values.astype(float), log, final, continuous_cols, pins


@pytest.mark.parametrize("tag,csv_name", [("e2e_hospital", "hospital.csv"), ("e2e_wine_MNAR", "wine.csv"),
                                          ("e2e_adult", "adult.csv"), ("e2e_spam_MNAR", "spam.csv")])
def test_end_to_end(tmp_path, tag, csv_name):
    """hospital: 9 columns, 7 categorical (one-hot heavy), MCAR 0.2, fallback validation split;
    wine: 13 numeric columns, MNAR (BASELINE.json config 2 style), 142 rows;
    spam: BASELINE.json configs[1] on the largest numeric UCI table of the reference -- 3 680 rows x 57 numeric columns, MNAR,
    k = 184 (tests/golden/make_e2e_adult.py 100 spam MNAR);
    adult: BASELINE.json configs[0] at its full size -- 24 118 rows after the train split, 14 columns (8 categorical,
    D = 118), MCAR 0.2, top_k 0.05 (k = 1 205: the window selection on a real table), Learner on the income label, FD
    plug-in on education / education-num, fallback validation split (tests/golden/make_e2e_adult.py).  About 1 % of
    adult's attention-filled numeric cells are near-ties on which the reference's fp32 rounding decides (adjudicated in
    test_attention_cells_adjudicated), so its diffusion stage is run on the reference's attention table: the
    trajectory pins and scores then compare the training / sampling arithmetic on identical inputs."""
    z, data_m, attn, log, final, con, pins = _run_pipeline(tmp_path, tag, csv_name, attn_from_reference=tag in ("e2e_adult", "e2e_spam_MNAR"))
    assert np.array_equal(np.asarray(data_m), z["data_m"])
    # attention fill: numeric cells to 1e-4 relative; categorical cells (arg-max of a softmax-weighted
    # vote) may flip only on near-ties
    ref = z["attn_coded"]
    num = np.zeros(ref.shape[1], dtype=bool); num[list(con)] = True
    if num.any():
        # near-tie slack: this table is one-hot heavy, kept sets hold hundreds of tied
        # logits, and the reference thresholds fp32 probabilities: a flipped near-tie member moves a cell
        # by <= 1/k of the value spread.  Everything else agrees to 1e-4 relative.
        # (spam: word frequencies, most cells 0 and fills of the order 1e-8: relative to the larger of |value| and the
        # column's standard deviation there, a relative error of a value that is zero means nothing)
        floor = ref[:, num].std(0, keepdims=True) if tag == "e2e_spam_MNAR" else 1e-6
        rel = np.abs(attn[:, num] - ref[:, num]) / np.maximum(np.abs(ref[:, num]), floor)
        # adult: 19 000 valid keys inside a logit range of 0.002 ... 0.03 put about ten keys within 1e-6 of EVERY k-th
        # logit, and the reference's kept set among them follows its fp32 rounding; test_attention_cells_adjudicated
        # shows in fp64 that every cell counted here is such a near-tie and that this build's value is the exact one
        frac_max, rel_max = (0.012, 0.1) if tag == "e2e_adult" else (0.012, 0.5) if tag == "e2e_spam_MNAR" else (0.005, 2e-3)
        assert (rel > 1e-4).mean() < frac_max, f"{(rel > 1e-4).mean():.4%} of numeric attention cells beyond 1e-4"
        assert rel.max() < rel_max, rel.max()
    mism = (attn[:, ~num] != ref[:, ~num]).mean() if (~num).any() else 0.0
    assert mism < 0.002, f"categorical attention cells differing from the reference: {mism:.4%}"
    scores = z["scores"]
    print("scores", log, "reference", scores.tolist())
    assert len(log) == len(scores)
    print("trajectory pins (max abs weight difference)", pins)
    # epochs 1 and 2 pin the arithmetic of a training step; by epoch 10 the fp32 summation order of the GEMMs has been
    # amplified (the reference's own runs at different thread counts are 2e-5 apart after TWO epochs,
    # e2e_hospital_spread.json), so that pin only guards against a gross divergence
    if "valid_rows_differing" in pins:
        # hospital: measured 28 of 270 entries; torch's own CPU path moves 11 - 13 between 1 and 8 threads.  adult (5 830
        # rows selected by 1 000 full-table Adam steps through dropout): measured 1 841; torch's own CPU path differs from
        # ITSELF by 1 685 - 1 781 entries between 1, 2 and 8 threads (tools/valid_index_sensitivity.py adult) -- the
        # selection is not a reproducible quantity there, the bound only guards against a different procedure
        # spam (736 rows selected): measured 94; torch against itself 46 - 48
        assert pins.pop("valid_rows_differing") <= {"e2e_adult": 2500, "e2e_spam_MNAR": 150}.get(tag, 40)
    for nm, d in pins.items():
        tol = 1e-7 if nm.endswith("ep1") else 1e-6 if nm.endswith("ep2") else 5e-4
        assert d < tol, (nm, d)
    if tag == "e2e_hospital":
        # the reference's own distribution on this run (see the module docstring): mean and standard deviation per pass
        sp = json.load(open(os.path.join(GOLD, "e2e_hospital_spread.json")))
        assert len(sp["runs"]) >= 8 and max(sp["armse_std_per_pass"] + sp["amae_std_per_pass"]) < 1.5e-3
        for i, (ep, r, m) in enumerate(log):
            assert abs(r - sp["armse_mean_per_pass"][i]) <= 3 * sp["armse_std_per_pass"][i], (ep, r, sp["armse_mean_per_pass"][i])
            assert abs(m - sp["amae_mean_per_pass"][i]) <= 3 * sp["amae_std_per_pass"][i], (ep, m, sp["amae_mean_per_pass"][i])
        # the returned table is the best pass's: the same bound against the reference's final scores
        assert abs(final[0] - float(z["armse"])) <= 4 * max(sp["armse_std_per_pass"])
        assert abs(final[1] - float(z["amae"])) <= 4 * max(sp["amae_std_per_pass"])
    elif tag == "e2e_adult":
        # the reference's own two runs (8 and 4 intra-op threads, tests/golden/e2e_adult_spread.json) are 6e-5 apart in both
        # scores after 100 epochs; "identical to 3 decimals" is asserted as |difference| <= 5e-4
        sp = json.load(open(os.path.join(GOLD, "e2e_adult_spread.json")))
        assert max(abs(a_ - b_) for a_, b_ in zip(sp["runs"][0]["final"], sp["runs"][1]["final"])) < 2e-4
        for (ep, r, m), (rr, mm) in zip(log, scores):
            assert abs(r - rr) <= 5e-4 and abs(m - mm) <= 5e-4, (ep, r, rr, m, mm)
        assert abs(final[0] - float(z["armse"])) <= 5e-4 and abs(final[1] - float(z["amae"])) <= 5e-4
    else:
        # all-numeric table, no arg-max to flip: scores agree far inside 3 decimals
        for (ep, r, m), (rr, mm) in zip(log, scores):
            assert abs(r - rr) <= 2e-4 and abs(m - mm) <= 2e-4, (ep, r, rr, m, mm)
        assert abs(final[0] - float(z["armse"])) <= 2e-4 and abs(final[1] - float(z["amae"])) <= 2e-4


def test_imputation_path_on_reference_weights(tmp_path):
    """The imputation hot path in isolation: sampler -> mask-merge -> decode -> score, on the weights the REFERENCE had
    trained when it sampled (epochs 100 and 200 of the hospital run) and on the random draws it made
    (tests/golden/e2e_hospital_passes.npz: state_dicts, the state of torch's CPU generator right before the pass, the
    pass's outputs; the first pass also holds the draws themselves).  Asserted: x0_hat, discriminator_out and the merged
    code within 2e-5 of the reference's; every decoded cell identical except arg-max flips between two categories
    whose reference probabilities differ by < 1e-4 (enumerated); ARMSE / AMAE identical to 3 decimals."""
    from ddgani_amd.model import Diffusion as D_, Discriminator_model as DM
    from ddgani_amd.utils import util
    from ddgani_amd.utils.data_loader import data_loader, set_label, value_loader
    from ddgani_amd import ops
    from util_golden import state_dict
    dev = torch.device("cuda:0")
    z, zp = load("e2e_hospital"), load("e2e_hospital_passes")
    param = json.loads(str(z["param_json"]))
    path = str(tmp_path) + "/"
    label_file, _ = set_label(os.path.join(GOLD, "hospital.csv"), path, "True")
    np.random.seed(0); random.seed(0); torch.manual_seed(0)
    (ori_data, miss_data, data_m, _, _, _, continuous_cols, _, _, all_data) = data_loader(
        label_file, float(z["miss_rate"]), param["categorical_cols"], "True", "MCAR", int(z["seed"]))
    values, _, value_cat = value_loader(label_file, continuous_cols, "True")
    _, enc = util.categorical_to_code(all_data, value_cat, enc=None)
    assert np.array_equal(np.asarray(data_m), z["data_m"])
    # the table the reference fed to the diffusion stage: its own attention-filled table (ordinal codes -> labels)
    attn = pd.DataFrame(z["attn_coded"], columns=values)
    attn[value_cat] = enc.inverse_transform(attn[value_cat])
    coded, _ = util.categorical_to_code(attn.copy(), value_cat, enc)
    coded.columns = list(range(coded.shape[1]))
    fields, feed = util.Data_convert(coded, param["model_name"], continuous_cols)
    x = torch.tensor(feed.values, dtype=torch.float)
    assert np.array_equal(x.numpy(), zp["p0__x"])                 # host encode == the reference's, bit for bit
    x = x.to(dev)
    N, D = x.shape
    T, C = int(param["T"]), len(fields)
    M = util.get_M_by_data_m(data_m, fields, dev)
    DS = D_.Diffusion_setting(T, D, dev)
    widths = (2 * D, D, 2 * D)
    is_cat = np.array([f.dim() > 1 for f in fields])
    off = np.concatenate([[0], np.cumsum([f.dim() for f in fields])])
    for i in range(int(zp["n_passes"])):
        G = D_.Generator_x0(D, 2 * D, D, T + 1, fields)
        G.load_state_dict(state_dict(zp, f"p{i}__G__"))
        Dn = DM.Discriminator_noise_x(2 * D, D // 2, C, T + 1)
        Dn.load_state_dict(state_dict(zp, f"p{i}__Dn__"))
        G.to(dev).train(); Dn.to(dev).train()
        torch.set_rng_state(torch.tensor(zp[f"p{i}__rng_state"]))
        rand, eps2 = D_.host_sampler_randomness(D_.HostRandom(), T, N, D, widths, dev)
        if f"p{i}__eps_T" in zp.files:                            # the replay reproduces the reference's draws
            assert np.array_equal(rand.eps_T.cpu().numpy(), zp[f"p{i}__eps_T"])
            assert np.array_equal(eps2.cpu().numpy(), zp[f"p{i}__eps2"])
            for j, ts in enumerate(range(T, 1, -1)):
                if ts > 2:
                    assert np.array_equal(rand.zs[j].cpu().numpy(), zp[f"p{i}__z_t{ts}"])
                for l in range(3):
                    bits = np.unpackbits(zp[f"p{i}__drop_t{ts}_{l}"], axis=1)[:, :widths[l]]
                    assert np.array_equal(rand.drops[j][l].cpu().numpy(), bits)
        x0_hat, d_out = D_.reverse_diffusion_G_D_sample(G, Dn, DS, T, N, D, dev, x, M, randomness=rand, eps2=eps2)
        np.testing.assert_allclose(x0_hat.cpu().numpy(), zp[f"p{i}__x0_hat"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(d_out.cpu().numpy(), zp[f"p{i}__d_out"], rtol=0, atol=2e-5)
        code = ops.mask_merge(x, M, x0_hat, G.layout)
        ref_code = zp[f"p{i}__code"]
        np.testing.assert_allclose(code.cpu().numpy(), ref_code, rtol=0, atol=2e-5)
        dec = pd.DataFrame(util.reconvert_data(code, fields, value_cat, values, attn.copy(), data_m, enc))
        dec.columns = values
        score = util.errorLoss(dec, ori_data, data_m, value_cat, continuous_cols, enc)
        dec[value_cat] = enc.transform(dec[value_cat])
        mine, ref = dec.values.astype(float), zp[f"p{i}__decoded"]
        flips = []
        for c in range(C):
            if is_cat[c]:
                for r in np.nonzero(mine[:, c] != ref[:, c])[0]:
                    blk = np.sort(ref_code[r, off[c]:off[c + 1]])
                    flips.append((int(r), c, float(blk[-1] - blk[-2])))
            else:
                np.testing.assert_allclose(mine[:, c], ref[:, c], rtol=1e-4, atol=1e-6)
        print(f"pass {i}: arg-max near-tie flips {flips}; scores {score} reference {zp[f'p{i}__scores'].tolist()}")
        assert all(gap < 1e-4 for _, _, gap in flips), flips
        assert len(flips) <= 2
        rr, mm = zp[f"p{i}__scores"]
        assert abs(score[0] - rr) < 5e-4 and abs(score[1] - mm) < 5e-4, (score, rr, mm)
        assert round(score[0], 3) == round(float(rr), 3) or abs(score[0] - rr) < 1e-5


@pytest.mark.parametrize("tag,csv_name", [("e2e_hospital", "hospital.csv"), ("e2e_adult", "adult.csv")])
def test_end_to_end_throughput_mode(tmp_path, tag, csv_name):
    """The same pipeline with on-chip random numbers and the training steps replayed as HIP graphs (rng="chip": what
    a production run uses), on hospital and on adult at its full size (device encode / decode, window selection, fallback
    discriminator on the device).  Not a replay of the reference's streams, so the scores are compared statistically:
    after 100 epochs within 0.02 of the reference's ARMSE / AMAE on this run (its own thread-count spread is 0.003)."""
    z, data_m, attn, log, final, con, pins = _run_pipeline(tmp_path, tag, csv_name, rng="chip", epochs=100)
    assert len(log) == 1 and log[0][0] == 100
    print("throughput mode", tag, log, "reference", z["scores"].tolist())
    assert abs(log[0][1] - float(z["scores"][0][0])) < 0.02, log
    assert abs(log[0][2] - float(z["scores"][0][1])) < 0.02, log
    from ddgani_amd import ops
    # the library no longer holds the trainer's stream-offset counter: two plain calls draw the same noise
    x0 = torch.rand(64, 8, device="cuda:0")
    assert torch.equal(ops.forward_noise(x0, 3, 6, seed=1, stream_id=5), ops.forward_noise(x0, 3, 6, seed=1, stream_id=5))


@pytest.mark.parametrize("tag,csv_name,shape,col_frac", [("e2e_adult", "adult.csv", (24118, 1205, 118), 0.07),
                                                         ("e2e_spam_MNAR", "spam.csv", (3680, 184, 57), 0.25)])
def test_attention_cells_adjudicated(tmp_path, tag, csv_name, shape, col_frac):
    """The attention fill of the reference's ``adult`` table at its full size (24 118 rows, k = 1 205, the window
    selection) and of its ``spam`` table under MNAR (3 680 rows, 57 sparse numeric columns, k = 184, the radix path)
    against the tables the reference itself produced (tests/golden/e2e_adult.npz, e2e_spam_MNAR.npz), cell by cell:
     * every numeric cell that differs by more than 1e-4 relative, and every categorical cell that differs, is
       recomputed in fp64 on the device from the same encoded inputs: its k-th and (k+1)-th largest logits are closer
       than 1e-6 (the north star's tie exemption: which of the tied keys the reference keeps follows the rounding of its
       fp32 probabilities), and this build's encoded value equals the fp64 closed form to 2e-6;
     * a sample of the agreeing cells equals the fp64 closed form as well;
     * no more than ``col_frac`` of a column's missing cells differ at all, and at most 0.1 % of the categorical cells."""
    from ddgani_amd import ops
    from ddgani_amd.BaseLine import Mean
    from ddgani_amd.utils import util
    from ddgani_amd.utils.data_loader import data_loader, set_label, value_loader
    dev = torch.device("cuda:0")
    z = load(tag)
    param = json.loads(str(z["param_json"]))
    path = str(tmp_path) + "/"
    label_file, _ = set_label(os.path.join(GOLD, csv_name), path, "True")
    miss_rate, seed = float(z["miss_rate"]), int(z["seed"])
    np.random.seed(0); random.seed(0); torch.manual_seed(0)
    (_, miss_data, data_m, _, _, _, con, nan_data, _, all_data) = data_loader(
        label_file, miss_rate, param["categorical_cols"], "True", str(z["miss_type"]), seed)
    values, _, value_cat = value_loader(label_file, con, "True")
    _, enc = util.categorical_to_code(all_data, value_cat, enc=None)
    dm = np.asarray(data_m)
    assert np.array_equal(dm, z["data_m"])
    aid = pd.DataFrame(Mean.fill_data_mean(nan_data, con), columns=values)
    corr = z["corr"]
    df, code = util.init_attn_2(corr, aid.copy(), data_m, param["categorical_cols"], enc, value_cat, dev, param["top_k"])
    a = df.copy()                                     # (a copy: reconvert_data ordinal-encodes the frame it is given)
    if value_cat:
        a[value_cat] = enc.transform(a[value_cat])
    mine, ref = a.values.astype(float), z["attn_coded"]
    # the encoded inputs of that call, for the fp64 closed form
    fields, X, M, W, widths = util.attention_inputs(corr, aid, data_m, param["categorical_cols"], enc, value_cat, device=dev)
    lay = ops.FieldLayout(widths, dev)
    off = lay.offsets
    N = X.shape[0]
    k = util.topk_count(N, param["top_k"])
    assert (N, k, lay.D) == shape
    X64, W64 = (X * M).double(), W.double()
    m8 = torch.tensor(dm != 0, dtype=torch.uint8, device=dev)

    def closed_form(j, c):
        kk = torch.nonzero(m8[:, c] == 1).flatten()
        kk = kk[kk != j]
        s_ = X64[kk] @ (W64[c] * X64[j])
        if s_.numel() <= k:                                   # fewer valid keys than k: all are kept, no k-th gap
            top = torch.cat([torch.sort(s_, descending=True).values, s_.new_full((k + 1 - s_.numel(),), -1e30)])
        else:
            top = torch.topk(s_, k + 1).values
        keep = s_ >= top[k - 1]
        w = torch.exp(s_ - s_.max())
        xv = X64[kk][:, off[c]:off[c + 1]]
        return float(top[k - 1] - top[k]), (w[keep, None] * xv[keep]).sum(0) / w[keep].sum()

    rng = np.random.default_rng(5)
    n_adj = 0
    for c in range(len(values)):
        miss = dm[:, c] == 0
        if c in con:
            floor = ref[:, c].std() if tag == "e2e_spam_MNAR" else 1e-6      # (see test_end_to_end)
            rel = np.abs(mine[:, c] - ref[:, c]) / np.maximum(np.abs(ref[:, c]), floor)
            bad = np.where((rel > 1e-4) & miss)[0]
            assert len(bad) <= col_frac * miss.sum() + 1, (values[c], len(bad), int(miss.sum()))
        else:
            bad = np.where((mine[:, c] != ref[:, c]) & miss)[0]
            assert len(bad) <= 1e-3 * miss.sum() + 1, (values[c], len(bad))
        assert np.array_equal(mine[~miss, c], ref[~miss, c])              # observed cells untouched
        cand = np.setdiff1d(np.where(miss)[0], bad)
        good = rng.choice(cand, min(3, len(cand)), replace=False)
        for j in bad:
            gap, fill = closed_form(int(j), c)
            assert gap <= 1e-6, (values[c], int(j), gap, mine[j, c], ref[j, c])
            assert float((code[j, off[c]:off[c + 1]].double() - fill).abs().max()) <= 2e-6, (values[c], int(j))
            n_adj += 1
        for j in good:
            _, fill = closed_form(int(j), c)
            assert float((code[j, off[c]:off[c + 1]].double() - fill).abs().max()) <= 2e-6, (values[c], int(j))
    print("cells adjudicated in fp64:", n_adj)
    assert n_adj > 0
