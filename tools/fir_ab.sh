#!/bin/bash
cd "$(dirname "$0")/.."
L=gpurun_out/pfb_ab.log
: > $L
run() { echo "## $*" >> $L; env "$@" timeout 600 python tools/pfb_probe.py 2>&1 | tail -1 >> $L; }
run NB=1600 BLOCK=16777216
run NB=1600 BLOCK=16777216 RCF_PFB5_STAG=1
run NB=1600 BLOCK=4194304
run NB=1600 BLOCK=4194304 RCF_PFB5_STAG=1
cat $L
