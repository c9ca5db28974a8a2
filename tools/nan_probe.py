"""Diagnostic for the throughput-mode NaN of round 2: the hospital pipeline in rng="chip" mode with
the OLD feed path (DDG_GRAPH_FEED=direct), recording -- without any extra host synchronisation -- what every training
step of epochs REC_FROM..REC_TO consumed and produced, so that a bad run can be diffed against a good one.

    python tools/nan_probe.py OUT.pt          # one run; exit code 0, prints GOOD / BAD <first non-finite record>
    python tools/nan_probe.py --diff A.pt B.pt

Not product code and not a test: it monkeypatches ddgani_amd.model.Diffusion from the outside."""
import json, os, random, sys, tempfile, types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REC_FROM, REC_TO, STOP_AFTER = int(os.environ.get("REC_FROM", 15)), int(os.environ.get("REC_TO", 20)), int(os.environ.get("STOP_AFTER", 21))


def csum(t):
    """device-side fingerprint of a tensor: (sum, sum |x|, non-finite count) in fp64 -- no host sync"""
    d = t.detach().double().flatten()
    return torch.stack([d.sum(), d.abs().sum(), (~torch.isfinite(d)).sum().double()])


class Stop(Exception):
    pass


def run(out_path):
    from util_golden import GOLD, load
    from ddgani_amd import ops
    from ddgani_amd.BaseLine import DiffGANI
    from ddgani_amd.model import Diffusion as D_
    from ddgani_amd.utils import util
    from ddgani_amd.utils.data_loader import data_loader, set_label, value_loader

    rec = []                      # (tag, epoch, step, {name: device tensor})
    state = {"epoch": 0, "on": False, "gstep": 0, "dstep": 0}

    orig_gen_epoch, orig_disc_epoch = D_.train_one_epoch_Generator, D_.train_one_epoch_discriminator_noise_x
    orig_tgg, orig_adam = ops.train_gen_grads, ops.ParamGrads.adam_step
    orig_gs_step = D_.GraphedSteps.step

    def gen_epoch(model, disc, DS, x, optimizer, *a, **k):
        epoch = a[2]
        state.update(epoch=epoch, on=REC_FROM <= epoch <= REC_TO, gstep=0, dstep=0, optG=optimizer)
        if epoch > STOP_AFTER:
            raise Stop()
        return orig_gen_epoch(model, disc, DS, x, optimizer, *a, **k)

    def disc_epoch(gen, disc, DS, x, optimizer, *a, **k):
        state.update(optD=optimizer)
        return orig_disc_epoch(gen, disc, DS, x, optimizer, *a, **k)

    def tgg(PG, PD, grads, x0, M, m, t, *a, **k):
        r = orig_tgg(PG, PD, grads, x0, M, m, t, *a, **k)
        if state["on"] and not torch.cuda.is_current_stream_capturing():
            d = {"x0": csum(x0), "M": csum(M), "m": csum(m), "t": t.clone(), "loss": r[0].clone(), "flat": csum(grads.flat)}
            if r[1] is not None:
                d["pred"] = csum(r[1])
            for kk, v in PG.t.items():
                d["PG." + kk] = csum(v)
            for kk, v in PD.t.items():
                d["PD." + kk] = csum(v)
            rec.append(("G", state["epoch"], state["gstep"], d))
        state["gstep"] += 1
        return r

    def gs_step(self, key, fn, rnd):
        slot = self.slot
        l = orig_gs_step(self, key, fn, rnd)
        if state["on"]:
            buf = state["optD"].buf if key[0] == "D" else state["optG"].buf
            d = {"feed_dev": self.feed.clone(), "feed_host": torch.from_numpy(self.stage_np[slot].copy()),
                 "loss": l.clone(), "flat": csum(buf.flat), "counter": self.counter.clone()}
            P = state["optD"].P if key[0] == "D" else state["optG"].P
            for kk, v in P.t.items():
                d["P." + kk] = csum(v)
            rec.append((key[0] + "graph", state["epoch"], state["dstep"], d))
        state["dstep"] += 1
        return l

    D_.train_one_epoch_Generator, D_.train_one_epoch_discriminator_noise_x = gen_epoch, disc_epoch
    ops.train_gen_grads = tgg
    D_.GraphedSteps.step = gs_step

    z = load("e2e_hospital")
    param = json.loads(str(z["param_json"]))
    param["epochs"] = 100
    path = tempfile.mkdtemp() + "/"
    label_file, label_num = set_label(os.path.join(GOLD, "hospital.csv"), path, "True")
    miss_rate, seed = float(z["miss_rate"]), int(z["seed"])
    args = types.SimpleNamespace(UseAttention="True", UseLearner="True", UseFD=os.environ.get("USE_FD", "True"))
    np.random.seed(0); random.seed(0); torch.manual_seed(0)
    (ori_data, miss_data, data_m, label_data, _, _, con, nan_data, _, all_data) = data_loader(
        label_file, miss_rate, param["categorical_cols"], "True", "MCAR", seed)
    values, _, value_cat = value_loader(label_file, con, "True")
    miss_data.columns = values
    miss_data.to_csv(path + f"miss_data_{miss_rate}.csv", index=None)
    _, enc = util.categorical_to_code(all_data, value_cat, enc=None)
    try:
        DiffGANI.get_Diff_acc_RMSE(nan_data, path, miss_rate, miss_data, enc, data_m, param["categorical_cols"], con,
                                   value_cat, torch.device("cuda:0"), param, label_data, values, ori_data, label_num,
                                   args, rng="chip", log=[])
    except Stop:
        pass
    torch.cuda.synchronize()
    host = [(tag, ep, st, {k: v.cpu() for k, v in d.items()}) for tag, ep, st, d in rec]
    torch.save(host, out_path)
    bad = None
    for tag, ep, st, d in host:
        for k, v in d.items():
            nf = (not torch.isfinite(v).all()) if v.dtype.is_floating_point and v.numel() != 3 else (v.numel() == 3 and v[2] != 0)
            if nf:
                bad = (tag, ep, st, k)
                break
        if bad:
            break
    mism = [(tag, ep, st) for tag, ep, st, d in host if "feed_dev" in d and not torch.equal(d["feed_dev"], d["feed_host"])]
    print("RESULT", "BAD %s" % (bad,) if bad else "GOOD", "records", len(host), "feed mismatches", mism[:6], flush=True)


def diff(a, b):
    A, B = torch.load(a), torch.load(b)
    for (ta, ea, sa, da), (tb, eb, sb, db) in zip(A, B):
        assert (ta, ea, sa) == (tb, eb, sb)
        for k in da:
            if not torch.equal(da[k], db[k]) and not (torch.isnan(da[k]).all() and torch.isnan(db[k]).all()):
                print("FIRST DIFFERENCE at", ta, "epoch", ea, "step", sa, "field", k)
                print("  A:", da[k].flatten()[:12].tolist())
                print("  B:", db[k].flatten()[:12].tolist())
                # everything that differs in this record
                print("  all differing fields of the record:", [kk for kk in da if not torch.equal(da[kk], db[kk])])
                return
    print("NO DIFFERENCE in", len(A), "records")


if __name__ == "__main__":
    if sys.argv[1] == "--diff":
        diff(sys.argv[2], sys.argv[3])
    else:
        run(sys.argv[1])
