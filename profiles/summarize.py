#!/usr/bin/env python3
"""Digest a rocprofv3 output directory (kernel-trace --stats pass + the two --pmc passes that
the round's gpurun command produced) into profiles/<tag>_*.  Usage:
    python profiles/summarize.py gpurun_out/prof_r1a r1a "config1:lut:1000000"
Traffic follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE come from separate
--pmc passes, are in KiB, and on gfx950 FETCH_SIZE is doubled (it tallies 128-B requests as
64 B for wide coalesced reads; our access mix is narrower, so the doubled figure is an upper
bound -- both are recorded)."""
import collections
import csv
import json
import os
import re
import shutil
import sys

src, tag, key = sys.argv[1], sys.argv[2], sys.argv[3]
here = os.path.dirname(os.path.abspath(__file__))
shutil.copy(os.path.join(src, "kt_kernel_stats.csv"), os.path.join(here, f"{tag}_kernel_stats.csv"))
pmc = {}
for name, f in (("FETCH_SIZE", "pmc_fetch_counter_collection.csv"), ("WRITE_SIZE", "pmc_write_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(os.path.join(src, f))):
        if row["Counter_Name"] == name:
            agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        m = re.search(r"(match_stats\w*kernel<[^>]*>|traverse_kernel<[^>]*>|interval_kernel<[^>]*>|interval_kernel|compact_\w+|sa_interval_kernel|seed_lookup_kernel<\d>)", k)
        if m:
            pmc.setdefault(m.group(1), {})[name + "_KiB_per_launch"] = sum(v) / len(v)
            pmc[m.group(1)]["launches"] = len(v)
for k, d in pmc.items():
    f, w = d.get("FETCH_SIZE_KiB_per_launch", 0.0), d.get("WRITE_SIZE_KiB_per_launch", 0.0)
    d["hbm_bytes_per_launch_raw"] = (f + w) * 1024
    d["hbm_bytes_per_launch_gfx950_corrected"] = (2 * f + w) * 1024
stats = list(csv.DictReader(open(os.path.join(src, "kt_kernel_stats.csv"))))
out = {"tag": tag, "workload": key, "kernel_stats": [
    {"name": r["Name"][:110], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])}
    for r in stats if "genie" in r["Name"]], "pmc": pmc}
json.dump(out, open(os.path.join(here, f"{tag}_summary.json"), "w"), indent=1)
# the value bench.py reports as roofline.traffic (dominant kernel, corrected bytes per launch)
tpath = os.path.join(here, "pmc_traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
dom = [k for k in pmc if k.startswith("match_stats")]
if dom:
    traffic[key] = pmc[dom[0]]["hbm_bytes_per_launch_gfx950_corrected"]
json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
