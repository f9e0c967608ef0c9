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

# --timeline [regex]: every launch of the matching kernels with its start and end (ms from the trace's first kernel), ordered by start --
# which launches overlap, and for how long, is read off directly (launches shorter than 0.05 ms are counted, not listed)
if "--timeline" in sys.argv:
    i = sys.argv.index("--timeline")
    pat = re.compile(sys.argv[i + 1] if len(sys.argv) > i + 1 else r"wavefront|hybrid_search")
    ev = []
    for q, lst in byq.items():
        for s, e, n in lst:
            if pat.search(n):
                ev.append((s, e, n, q))
    ev.sort()
    short = defaultdict(int)
    print("\nstart_ms    end_ms      ms        queue  kernel")
    for s, e, n, q in ev:
        if (e - s) < 50000:
            short[n] += 1
            continue
        print("%9.2f  %9.2f  %8.2f  %5s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
    for n, c in short.items():
        print("(%d launches of %s under 0.05 ms not listed)" % (c, n))
    # how many launches of each kernel are in flight on average over the span they cover
    for name in sorted(set(n for _, _, n, _ in ev)):
        sel = [(s, e) for s, e, n, _ in ev if n == name and e - s >= 50000]
        if sel:
            a, b = min(s for s, _ in sel), max(e for _, e in sel)
            print("%-24s launches in flight over its span of %.1f ms: %.2f" % (name, (b - a) / 1e6, sum(e - s for s, e in sel) / max(1, b - a)))
