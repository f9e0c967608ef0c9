#!/bin/bash
# Average active lanes per VALU instruction (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU) of the bench kernels.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_lanes
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/a -o run -- python3 bench.py ${PMC_BENCH_ARGS:---steps 8 --warmup 8} --check-poses 1024 --no-cpu-baseline > $OUT/a.log 2>&1
python3 - <<PY
import csv, glob, re
agg = {}
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        if not m: continue
        d = agg.setdefault(m.group(1), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, d in agg.items():
    if "SQ_THREAD_CYCLES_VALU" in d and d.get("SQ_ACTIVE_INST_VALU"):
        print("%-24s thread-cycles %.3g  active-inst-cycles %.3g  insts %.3g  -> lanes/inst %.1f" % (k, d["SQ_THREAD_CYCLES_VALU"], d["SQ_ACTIVE_INST_VALU"], d.get("SQ_INSTS_VALU", 0), d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"]))
PY
