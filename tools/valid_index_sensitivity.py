"""How sensitive is get_valid_data_index (reference utils/util.py:623-646) to fp32 summation order?  The stock torch CPU
path of the mirror (the reference's own op sequence) on the `hospital` run with 1, 2 and 8 intra-op threads, everything
else equal: measured 11 - 13 of 270 selected rows differ between thread counts (1 000 full-batch Adam steps amplify
last-bit differences; the rows near the 20 % cut-off swap).  tests/test_gpu_e2e.py bounds the device path accordingly."""
import json, os, random, sys, tempfile
import numpy as np, pandas as pd, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from util_golden import GOLD, load
from ddgani_amd.utils import util
from ddgani_amd.utils.data_loader import data_loader, set_label, value_loader
from ddgani_amd.model.Discriminator_model import D
TABLE = sys.argv[1] if len(sys.argv) > 1 else "hospital"      # hospital | adult (adult: 1 841 of the device path's entries
z = load("e2e_" + TABLE)                                       # (also: spam_MNAR)                                       # differ from the reference's, see the measured numbers below)
param = json.loads(str(z["param_json"]))
path = tempfile.mkdtemp() + "/"
label_file, _ = set_label(os.path.join(GOLD, TABLE.split("_")[0] + ".csv"), path, "True")
np.random.seed(0); random.seed(0); torch.manual_seed(0)
(ori_data, miss_data, data_m, _, _, _, con, _, _, all_data) = data_loader(label_file, float(z["miss_rate"]), param["categorical_cols"], "True", str(z["miss_type"]) if "miss_type" in z.files else "MCAR", int(z["seed"]))
values, _, value_cat = value_loader(label_file, con, "True")
_, enc = util.categorical_to_code(all_data, value_cat, enc=None)
attn = pd.DataFrame(z["attn_coded"], columns=values)
if value_cat:
    attn[value_cat] = enc.inverse_transform(attn[value_cat])
coded, _ = util.categorical_to_code(attn.copy(), value_cat, enc)
coded.columns = list(range(coded.shape[1]))
fields, feed = util.Data_convert(coded, param["model_name"], con)
x = torch.tensor(feed.values, dtype=torch.float)
Dm, C = x.shape[1], len(fields)
res = {}
for th in (1, 2, 8):
    torch.set_num_threads(th)
    torch.manual_seed(3047)           # model/Diffusion.py:555 precedes get_valid_data_index
    disc = D(Dm, Dm // 2, C)
    res[th] = np.asarray(util.get_valid_data_index(data_m, disc, x, "cpu"))
    print("threads", th, "selected", len(res[th]), "differs from the reference's set in", len(np.setxor1d(res[th], z["valid_index"])), flush=True)
print("1 vs 8 threads:", len(np.setxor1d(res[1], res[8])), " 1 vs 2:", len(np.setxor1d(res[1], res[2])))
