#!/bin/bash
# Regenerate profiles/ for a round on the GPU box:  tools/make_profiles.sh r02
#   1. bench.py JSON line (full run)
#   2. rocprofv3 --kernel-trace --stats of the same command (kernel summary CSV, all legs except the CPU baseline)
#   3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over the timed configuration -> profiles/pfb_traffic.json
#   4. counters of the matrix-core FIR bank at 4096 channels -> profiles/<R>_fir_mfma_pmc.json
#   5. counters + traffic of the 512-bin and the 1600-bin filterbank -> profiles/<R>_pfb512_traffic.json, <R>_pfb1600_pmc.json
# gpurun merges only gpurun_out/ back: run `python tools/collect_profiles.py <R>` locally afterwards.
# Counter passes never share a run with trace domains other than --kernel-trace.
set -u
R=${1:-r02}
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p profiles gpurun_out
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 20 --warmup 3"

python bench.py --steps 20 --warmup 3 > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err

rm -rf gpurun_out/${R}_trace
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_trace -- $CMD --no-cpu-baseline --sweep-max 16384 > /dev/null 2>&1)

for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${R}_pmc_$c
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/${R}_pmc_$c -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1)
done

tools/fir_pmc.sh ${R}_fir4096 C=4096 > /dev/null 2>&1
PROBE=tools/pfb_probe.py KERNEL=pfb_kernel tools/fir_pmc.sh ${R}_pfb512 NB=512 > /dev/null 2>&1
PROBE=tools/pfb_probe.py KERNEL=pfb5_kernel tools/fir_pmc.sh ${R}_pfb1600 NB=1600 BLOCK=33554432 > /dev/null 2>&1
echo done
