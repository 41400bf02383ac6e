#!/usr/bin/env python3
"""Digest one tools/profile_round.sh output directory (a kernel-trace --stats pass + one rocprofv3 --pmc pass per
counter set) into
    <dir>/summary.json                 per-kernel average duration and counters of this workload
    profiles/<tag>_summary.json        the same, tracked
    profiles/<tag>_kernel_stats.csv    rocprofv3's own --stats table
    profiles/pmc_counters.json         merged: what bench.py reports as roofline.traffic / step / binding
Usage:  python profiles/summarize.py gpurun_out/prof_<tag> <tag>

HBM traffic follows MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE come from separate --pmc passes, are in
KiB, and on gfx950 FETCH_SIZE counts half the bytes of a coalesced streaming read, so
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024           (an upper bound for our narrower access mix;
    hbm_bytes_raw = (FETCH_SIZE + WRITE_SIZE) * 1024            both are recorded)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("genie::(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in name:                       # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out)


bench = json.loads(open(os.path.join(src, "kt_bench.json")).read().strip().splitlines()[-1])
key = bench["config"]["counters_key"]
kernels = collections.defaultdict(dict)
for r in csv.DictReader(open(os.path.join(src, "kt_kernel_stats.csv"))):
    if "genie" in r["Name"]:
        kernels[short(r["Name"])].update(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), pct=float(r["Percentage"]))
for f in sorted(glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "genie" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            kernels[k][c] = sum(v) / len(v)                   # average per launch
for k, d in kernels.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
        d["hbm_bytes_raw"] = (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
out = {"tag": tag, "key": key, "workload": bench["config"]["workload"], "bench_line": {k: bench[k] for k in ("value", "ms_per_step", "n_gpus")},
       "xcds": 8, "simds": 1024, "kernels": kernels}
json.dump(out, open(os.path.join(src, "summary.json"), "w"), indent=1, sort_keys=True)
if os.path.isdir(here) and os.access(here, os.W_OK):
    shutil.copy(os.path.join(src, "kt_kernel_stats.csv"), os.path.join(here, f"{tag}_kernel_stats.csv"))
    json.dump(out, open(os.path.join(here, f"{tag}_summary.json"), "w"), indent=1, sort_keys=True)
    path = os.path.join(here, "pmc_counters.json")
    merged = json.load(open(path)) if os.path.exists(path) else {}
    merged[key] = {"tag": tag, "xcds": 8, "simds": 1024, "kernels": kernels}
    json.dump(merged, open(path, "w"), indent=1, sort_keys=True)
for k, d in sorted(kernels.items(), key=lambda kv: -kv[1].get("avg_ns", 0)):
    print(f"{k[:60]:60s} {d.get('avg_ns', 0) / 1e3:9.1f} us  hbm {d.get('hbm_bytes', 0) / 1e6:8.1f} MB  valu {d.get('SQ_INSTS_VALU', 0) / 1e6:7.1f} M  "
          f"l2rd {d.get('TCP_TCC_READ_REQ_sum', 0) / 1e6:7.1f} M")
