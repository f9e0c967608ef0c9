#!/usr/bin/env python3
"""Where a kernel's scratch traffic sits: attributes every scratch_load / scratch_store of one kernel in a
`hipcc -S -gline-tables-only` listing to the line of the KERNEL BODY it was inlined into (the outermost frame of the
.loc's inlined-at chain that lies in the given file), and sums by the phase ranges given on the command line.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -gline-tables-only -S --cuda-device-only \
          -I include -o /tmp/planner_g.s pathplanning_amd/csrc/pp_planner.hip
    python tools/isa_spill_map.py /tmp/planner_g.s k_hybrid_search_rowsILb1E pp_planner_rows.hpp --stamps=pathplanning_amd/csrc/pp_planner_rows.hpp
    (or explicit phases: name:first-last ...)
"""
import collections
import re
import sys


def phases_from_stamps(source_path):
    """The phases of k_hybrid_search_rows from its own ROWS_STAMP(n) markers (pp_planner_rows.hpp): [(name, first line, last line)].  A stamp closes the phase that
    precedes it; the children's sub-stamps (11, 12, 13, 4, 5, 6) are folded into one phase that ends at stamp 7."""
    src = open(source_path).read().split("\n")
    loop = next(i for i, l in enumerate(src) if l.strip() == "for (;;) {") + 1
    stamp = {}
    for i, l in enumerate(src):
        m = re.match(r"\s*ROWS_STAMP\((\d+)\)", l)
        if m:
            stamp[int(m.group(1))] = i + 1
    order = [("take", 0), ("set-aside", 1), ("pop+refill", 2), ("node", 3), ("children", 7), ("insertion", 8), ("node-records", 9), ("reeds-shepp", 10)]
    out, lo = [], loop
    for name, k in order:
        out.append((name, lo, stamp[k]))
        lo = stamp[k] + 1
    out.append(("lambdas+prologue", 1, loop - 1))
    return out


def spill_map(path, kernel, body, phases):
    """{phase: [instructions, scratch loads, scratch stores]} of one kernel in a `hipcc -S -gline-tables-only` listing"""
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kernel in l and ": " in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    cur = None
    per_line = collections.defaultdict(lambda: [0, 0, 0])
    inst = re.compile(r"(v_|s_|ds_|global_|scratch_|buffer_|flat_)")
    frame = re.compile(r"([\w./+-]+):(\d+):\d+")
    for l in lines[start:end]:
        s = l.strip()
        if s.startswith(".loc"):
            c = s.split(";", 1)[1] if ";" in s else ""
            inb = [int(n) for f, n in frame.findall(c) if f.split("/")[-1] == body and int(n) > 0]
            cur = inb[-1] if inb else cur
            continue
        if inst.match(s):
            e = per_line[cur]
            e[2] += 1
            if s.startswith("scratch_load"):
                e[0] += 1
            elif s.startswith("scratch_store"):
                e[1] += 1
    out = {}
    for name, lo, hi in phases:
        v = [0, 0, 0]
        for ln, e in per_line.items():
            if ln is not None and lo <= ln <= hi:
                v[0] += e[2]
                v[1] += e[0]
                v[2] += e[1]
        out[name] = v
    return out


def main():
    path, kernel, body = sys.argv[1], sys.argv[2], sys.argv[3]
    phases = []
    for a in sys.argv[4:]:
        if a.startswith("--stamps="):  # phases from the ROWS_STAMP markers of that source file
            phases = phases_from_stamps(a.split("=", 1)[1])
            continue
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
