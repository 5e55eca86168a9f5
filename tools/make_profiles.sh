#!/bin/bash
# Regenerate profiles/ for a round on the GPU box:  tools/make_profiles.sh r01
#   1. bench.py JSON line
#   2. rocprofv3 --kernel-trace --stats of the same command (kernel summary CSV)
#   3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/pfb_traffic.json
# gpurun merges only gpurun_out/ back: run `python tools/collect_profiles.py r01` locally afterwards.
# Counter passes never share a run with trace domains other than --kernel-trace.
set -u
R=${1:-r01}
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p profiles gpurun_out
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 20 --warmup 3"

python bench.py --steps 20 --warmup 3 > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err

(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_trace -- $CMD --no-cpu-baseline > /dev/null 2>&1)

for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/${R}_pmc_$c -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
done
python tools/collect_profiles.py "$R"
cat profiles/${R}_bench.json
head -12 profiles/${R}_bench_kernel_stats.csv
