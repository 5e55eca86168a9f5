#!/bin/bash
# PMC passes over one kernel of a probe script, on the GPU box:  tools/fir_pmc.sh <tag> [env...]
#   PROBE=tools/fir_probe.py (default) | tools/pfb_probe.py ...   KERNEL=fir_mfma (substring of the kernel name)
# Counter passes carry only --kernel-trace.  Summaries -> gpurun_out/<tag>_pmc.txt
cd "$(dirname "$0")/.."
ROOT=$PWD
TAG=$1; shift
export TMPDIR=/tmp
PROBE=${PROBE:-tools/fir_probe.py}
KERNEL=${KERNEL:-fir_mfma}
OUT=gpurun_out/${TAG}_pmc.txt
: > $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  d=$ROOT/gpurun_out/${TAG}_pmc_$i
  rm -rf $d
  (cd /tmp && env "$@" STEPS=3 timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python $ROOT/$PROBE > $d.log 2>&1)
  echo "## pass $i: $set" >> $OUT
  tail -1 $d.log >> $OUT
  python tools/pmc_summary.py $d $KERNEL >> $OUT
  # per-dispatch duration of the same kernel in this (profiled) pass
  python - "$d" "$KERNEL" >> $OUT <<'PY'
import csv, glob, os, sys
d = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if d:
    print("%-32s %16.1f  (n=%d)" % ("kernel_us(profiled pass)", sum(d) / len(d), len(d)))
PY
done
cat $OUT
