#!/usr/bin/env python3
"""Summarise one rocprofv3 --pmc pass of SQ counters per kernel as fractions of SQ_WAVE_CYCLES.
usage: summarize_sq.py <counter_collection.csv> <out.json> "<the command that was profiled>" """
import collections
import csv
import json
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        cnt[k] += 1
out = {"note": "rocprofv3 --pmc " + " ".join(sorted(next(iter(agg.values())))) + " -- " + sys.argv[3] +
       "; cycle counters as fractions of SQ_WAVE_CYCLES (quad-cycle units; SQ_VALU_MFMA_BUSY_CYCLES is in cycles: divide "
       "its fraction by 4), instruction counters per wave", "kernels": {}}
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    wc, wv = d["SQ_WAVE_CYCLES"], d["SQ_WAVES"]
    e = {"launches": cnt[k], "waves": wv, "wave_cycles": wc}
    for a, b in d.items():
        if a in ("SQ_WAVES", "SQ_WAVE_CYCLES"):
            continue
        e[a] = b / wv if a.startswith("SQ_INSTS") else b / wc
    out["kernels"][k] = e
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, e in out["kernels"].items():
    print(k[:50], {a: f"{b:.4g}" for a, b in e.items()})
