#!/bin/bash
# HBM-side traffic of the bench kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE),
# kernel-trace only, of `bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline`.
# Run on the GPU box from the repo root: bash tools/pmc_traffic.sh <tag>   -> gpurun_out/pmc_<tag>/traffic.json + CSVs
set -e
TAG=${1:-run}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
	rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o run -- python3 bench.py --mode lanes --steps 1 --warmup 0 --streams 1 --no-cpu-baseline > $OUT/$c.log 2>&1
	echo "pass $c done"
done
# L2 hit rate of the same launches: TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum), a third pass of its own
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/TCC_HIT_MISS -o run -- python3 bench.py --mode lanes --steps 1 --warmup 0 --streams 1 --no-cpu-baseline > $OUT/TCC_HIT_MISS.log 2>&1
echo "pass TCC_HIT_sum TCC_MISS_sum done"
python3 tools/pmc_summarize.py $OUT
