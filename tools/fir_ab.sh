#!/bin/bash
cd "$(dirname "$0")/.."
L=gpurun_out/pfb_ab.log
: > $L
run() { echo "## $*" >> $L; env "$@" timeout 600 python tools/pfb_probe.py 2>&1 | tail -2 >> $L; }
run NB=1600 BLOCK=16777216 RCF_PFB5_DBG=3
run NB=1600 BLOCK=16777216 RCF_PFB5_DBG=3 RCF_PFB5_LDSPAD=20000
run NB=1600 BLOCK=16777216 RCF_PFB5_DBG=3 RCF_PFB5_LDSPAD=40000
run NB=1600 BLOCK=33554432 RCF_PFB5_DBG=3
cat $L
