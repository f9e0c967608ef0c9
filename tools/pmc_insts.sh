#!/bin/bash
# Instruction mix of the bench kernels (separate --pmc pass, kernel-trace only).
set -e
TAG=${1:-insts}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/a -o run -- python3 bench.py ${PMC_BENCH_ARGS:---steps 1 --warmup 0 --streams 1} --no-cpu-baseline > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/b -o run -- python3 bench.py ${PMC_BENCH_ARGS:---steps 1 --warmup 0 --streams 1} --no-cpu-baseline > $OUT/b.log 2>&1 || true
python3 - <<PY
import csv, glob, re
agg = {}
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        if k not in ("k_hybrid_search", "k_wavefront", "k_hybrid_search_rows"): continue
        d = agg.setdefault(k, {})
        d[r["Counter_Name"]] = max(d.get(r["Counter_Name"], 0.0), float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s %.4g" % (c, v))
PY
