#!/usr/bin/env python3
"""Turns the two counter passes of tools/pmc_traffic.sh into traffic.json (bytes per launch, largest launch per kernel)."""
import csv
import glob
import json
import re
import sys

out = sys.argv[1]
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(1)
            v = float(r["Counter_Value"])
            d = raw.setdefault(k, {})
            d[c + "_KB"] = max(d.get(c + "_KB", 0.0), v)
res = {
    "_note": "HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, from two separate rocprofv3 --pmc passes "
    "(FETCH_SIZE, then WRITE_SIZE) of `bench.py --mode lanes --steps 1 --warmup 0 --streams 1` (4096 queries per launch; 2^26 poses for "
    "k_check_states); largest launch of each kernel. The factor 2 on FETCH_SIZE is the gfx950 correction of "
    "MI355X_MICROARCH.md (HBM section); it was calibrated on k_check_states_fused (reads only the 4.19 MB distance grid, "
    "once per XCD L2 = 33.5 MB expected, FETCH_SIZE*1024 reports 16.9 MB) and k_check_states (writes 1 B/pose, "
    "WRITE_SIZE*1024 = 1.007x that). Infinity-Cache hits are counted as traffic.",
    "_raw_KB": raw,
}
for k, d in raw.items():
    if "FETCH_SIZE_KB" in d and "WRITE_SIZE_KB" in d:
        res[k] = (2 * d["FETCH_SIZE_KB"] + d["WRITE_SIZE_KB"]) * 1024
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
for k, v in res.items():
    if not k.startswith("_"):
        print("%-24s %.3e B/launch" % (k, v))
