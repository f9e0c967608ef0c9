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
l2 = {}
for f in glob.glob("%s/TCC_HIT_MISS/**/*counter_collection.csv" % out, recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] not in ("TCC_HIT_sum", "TCC_MISS_sum"):
            continue
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        if not m:
            continue
        d = l2.setdefault(m.group(1), {}).setdefault(r.get("Dispatch_Id", "0"), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
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
# L2 hit rate per kernel: the dispatch with the most L2 accesses (the full-size launch)
res["l2_hit"] = {}
res["_l2_raw"] = {}
for k, per in l2.items():
    best = max(per.values(), key=lambda d: d.get("TCC_HIT_sum", 0.0) + d.get("TCC_MISS_sum", 0.0))
    h, mi = best.get("TCC_HIT_sum", 0.0), best.get("TCC_MISS_sum", 0.0)
    if h + mi > 0:
        res["l2_hit"][k] = h / (h + mi)
        res["_l2_raw"][k] = dict(TCC_HIT_sum=h, TCC_MISS_sum=mi)
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
for k, v in res.items():
    if not k.startswith("_") and k != "l2_hit":
        print("%-24s %.3e B/launch   L2 hit %s" % (k, v, ("%.3f" % res["l2_hit"][k]) if k in res["l2_hit"] else "-"))
