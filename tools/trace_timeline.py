#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV: per kernel name, number of launches, mean duration, and the mean gap between
the end of one launch and the start of the next on the same queue (how long launches waited)."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
byq = defaultdict(list)
for r in rows:
    m = re.search(r"(k_[a-z_0-9]+|fillBuffer|copyBuffer)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:20]
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
dur = defaultdict(list)
wait = defaultdict(list)
for q, ev in byq.items():
    ev.sort()
    for i, (s, e, n) in enumerate(ev):
        dur[n].append((e - s) / 1e6)
        if i:
            wait[n].append((s - ev[i - 1][1]) / 1e6)
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
print("span %.1f ms, %d queues" % ((t1 - t0) / 1e6, len(byq)))
for n in sorted(dur, key=lambda k: -sum(dur[k])):
    w = wait.get(n, [0])
    print("%-24s n=%4d  mean %.1f ms  max %.1f ms   start after previous on its queue: mean %.1f ms" % (n, len(dur[n]), sum(dur[n]) / len(dur[n]), max(dur[n]), sum(w) / max(1, len(w))))
