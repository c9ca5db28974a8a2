#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected in separate runs of the same
bench.py command, --steps 1 --warmup 0) into profiles/rNN/pmc_traffic_<workload>_rows<N>.json.
FETCH_SIZE is in KB and on gfx950 counts 64 B per 128-B request for wide coalesced reads
(MI355X_MICROARCH.md, HBM): fetch bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE KB is exact.
usage: summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <workload> <rows>"""
import collections
import csv
import json
import sys

fetch_csv, write_csv, out_json, workload, rows = sys.argv[1:6]
out = {"note": __doc__.split("usage:")[0].strip(), "workload": workload, "rows": int(rows), "kernels": {}}
for nm, path, fac in (("fetch", fetch_csv, 2048.0), ("write", write_csv, 1024.0)):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if not r["Kernel_Name"].startswith(("void k_", "k_")):
            continue
        a = agg[r["Kernel_Name"].split("(")[0]]
        a[0] += 1
        a[1] += float(r["Counter_Value"]) * fac
        a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (n, b, ns) in agg.items():
        d = out["kernels"].setdefault(k, {})
        d[f"{nm}_bytes_per_step"] = b
        d[f"{nm}_bytes_per_launch"] = b / n
        d[f"launches_per_step_{nm}_pass"] = n
        d[f"total_ns_{nm}_pass"] = ns
json.dump(out, open(out_json, "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("fetch_bytes_per_step", 0))[:8]:
    print(k[:44], {a: f"{b:.3e}" for a, b in v.items()})
