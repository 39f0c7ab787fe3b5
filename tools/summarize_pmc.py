#!/usr/bin/env python3
"""Collect per-kernel counter values from the rocprofv3 --pmc passes of tools/profile_pmc.sh into one JSON."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
kernels = ("k_seed_mems", "k_find_mems", "k_enum_jobs", "k_prefilter", "k_pack_queries", "k_place_inline", "k_lcp_kasai", "k_links")
out = {}
for d in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for k in kernels:
                if k in r["Kernel_Name"]:
                    e = out.setdefault(k, {}).setdefault(r["Counter_Name"], [])
                    e.append(float(r["Counter_Value"]))
                    dur = out[k].setdefault("_duration_ms_" + os.path.basename(d), [])
                    if len(dur) < len(e):
                        dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
res = {k: {c: (sum(v) / len(v)) for c, v in d.items()} for k, d in out.items()}
print(json.dumps(res, indent=1, sort_keys=True))
