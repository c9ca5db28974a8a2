#!/usr/bin/env python3
"""How far does the REFERENCE drift from itself on the end-to-end run when nothing but the fp32 summation
order of its CPU GEMMs changes?  Runs tests/golden/make_e2e.py (the reference on `hospital`, MCAR 0.2, seed 42,
200 epochs, all flags True, same seeds) with 1 to 8 intra-op threads and writes the score lists, the number
of decoded cells that differ from the 8-thread run, and the weight drift at the trajectory-pin epochs to
e2e_hospital_spread.json.  tests/test_gpu_e2e.py derives its end-to-end tolerance from this file.
Build container only (needs /root/reference)."""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def run(threads):
    out = f"/tmp/dec_{threads}.npy"
    env = dict(os.environ, DDG_REF_THREADS=str(threads), DDG_SCORES_ONLY="1", DDG_DECODED_OUT=out,
               OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    r = subprocess.run([sys.executable, os.path.join(HERE, "make_e2e.py"), "hospital", "200"], env=env,
                       capture_output=True, text=True, check=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("SCORES_JSON ")][0]
    d = json.loads(line[len("SCORES_JSON "):])
    return d, np.load(out), np.load(out + ".w.npy")


def main():
    runs = {}
    for n in (8, 1, 2, 3, 4, 5, 6, 7):
        runs[n] = run(n)
        print(n, runs[n][0], flush=True)
    base_d, base_dec, base_w = runs[8]
    res = {"what": "reference vs itself, intra-op threads (fp32 summation order of the CPU GEMMs) varied", "runs": []}
    for n, (d, dec, w) in runs.items():
        d["cells_differing_from_8_threads"] = [int((dec[i] != base_dec[i]).sum()) for i in range(dec.shape[0])]
        d["max_abs_G_temb_diff_vs_8_threads_ep1_2_10_100"] = [
            float(np.abs(a - b).max()) for a, b in zip(np.split(w, 4), np.split(base_w, 4))]
        res["runs"].append(d)
    sc = np.array([r["scores"] for r in res["runs"]])            # [run, pass, (armse, amae)]
    res["armse_spread_per_pass"] = (sc[:, :, 0].max(0) - sc[:, :, 0].min(0)).tolist()
    res["amae_spread_per_pass"] = (sc[:, :, 1].max(0) - sc[:, :, 1].min(0)).tolist()
    res["armse_mean_per_pass"] = sc[:, :, 0].mean(0).tolist()
    res["amae_mean_per_pass"] = sc[:, :, 1].mean(0).tolist()
    res["armse_std_per_pass"] = sc[:, :, 0].std(0, ddof=1).tolist()
    res["amae_std_per_pass"] = sc[:, :, 1].std(0, ddof=1).tolist()
    json.dump(res, open(os.path.join(HERE, "e2e_hospital_spread.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
