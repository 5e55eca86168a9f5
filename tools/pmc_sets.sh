#!/bin/bash
# ad-hoc counter sets over one kernel of a probe:  PROBE=... KERNEL=... tools/pmc_sets.sh <tag> "<set1>" "<set2>" ... -- [env...]
cd "$(dirname "$0")/.."
ROOT=$PWD
TAG=$1; shift
export TMPDIR=/tmp
PROBE=${PROBE:-tools/fir_probe.py}
KERNEL=${KERNEL:-fir_mfma}
SETS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do SETS+=("$1"); shift; done
shift
OUT=gpurun_out/${TAG}_pmc.txt
: > $OUT
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  d=$ROOT/gpurun_out/${TAG}_pmcx_$i
  rm -rf $d
  (cd /tmp && env "$@" STEPS=3 timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python $ROOT/$PROBE > $d.log 2>&1)
  echo "## $set" >> $OUT
  python tools/pmc_summary.py $d $KERNEL >> $OUT
done
cat $OUT
