#!/usr/bin/env python3
"""Where a kernel's scratch traffic sits: attributes every scratch_load / scratch_store of one kernel in a
`hipcc -S -gline-tables-only` listing to the line of the KERNEL BODY it was inlined into (the outermost frame of the
.loc's inlined-at chain that lies in the given file), and sums by the phase ranges given on the command line.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -gline-tables-only -S --cuda-device-only \
          -I include -o /tmp/planner_g.s pathplanning_amd/csrc/pp_planner.hip
    python tools/isa_spill_map.py /tmp/planner_g.s k_hybrid_search_rowsILb1E pp_planner_rows.hpp \
          take:392-575 idle:576-633 setaside:634-711 pop:712-719 node:720-797 children:798-922 insert:923-1008 write:1009-1034 rs:1035-1227
"""
import collections
import re
import sys


def main():
    path, kernel, body = sys.argv[1], sys.argv[2], sys.argv[3]
    phases = []
    for a in sys.argv[4:]:
        name, r = a.split(":")
        lo, hi = r.split("-")
        phases.append((name, int(lo), int(hi)))
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kernel in l and l.rstrip().endswith(":") or (l.startswith("_Z") and kernel in l and ": " in l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    cur = None
    per_line = collections.defaultdict(lambda: [0, 0, 0])  # loads, stores, all
    inst = re.compile(r"(v_|s_|ds_|global_|scratch_|buffer_|flat_)")
    frame = re.compile(r"([\w./+-]+):(\d+):\d+")
    for l in lines[start:end]:
        s = l.strip()
        if s.startswith(".loc"):
            c = s.split(";", 1)[1] if ";" in s else ""
            fr = [(f.split("/")[-1], int(n)) for f, n in frame.findall(c)]
            inb = [n for f, n in fr if f == body and n > 0]
            cur = inb[-1] if inb else cur  # outermost frame in the kernel's file; line 0 = compiler-generated: keep the last known
            continue
        if inst.match(s):
            e = per_line[cur]
            e[2] += 1
            if s.startswith("scratch_load"):
                e[0] += 1
            elif s.startswith("scratch_store"):
                e[1] += 1
    tot = [sum(v[i] for v in per_line.values()) for i in range(3)]
    print("kernel %s: %d instructions, %d scratch loads, %d scratch stores" % (kernel, tot[2], tot[0], tot[1]))
    if phases:
        print("%-12s %8s %8s %8s" % ("phase", "insts", "loads", "stores"))
        seen = set()
        for name, lo, hi in phases:
            v = [0, 0, 0]
            for ln, e in per_line.items():
                if ln is not None and lo <= ln <= hi:
                    seen.add(ln)
                    for i in range(3):
                        v[i] += e[i]
            print("%-12s %8d %8d %8d" % (name, v[2], v[0], v[1]))
        v = [0, 0, 0]
        for ln, e in per_line.items():
            if ln not in seen:
                for i in range(3):
                    v[i] += e[i]
        print("%-12s %8d %8d %8d" % ("(elsewhere)", v[2], v[0], v[1]))
    else:
        for ln, e in sorted(per_line.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:40]:
            print(ln, e)


if __name__ == "__main__":
    main()
