#!/bin/bash
cd "$(dirname "$0")/.."
L=gpurun_out/pfb_ab.log
: > $L
run() { echo "## $*" >> $L; env "$@" timeout 600 python tools/pfb_probe.py 2>&1 | tail -1 >> $L; }
run NB=256
run NB=512
run NB=1024
run NB=256 OS=2
run NB=1600 BLOCK=16777216
run NB=1600 BLOCK=16777216 RCF_PFB5_LDS=1
run NB=3200 BLOCK=16777216
N=1048576 FRAMES=200 python tools/scan_probe.py 2>&1 | tail -1 >> $L
N=16384 FRAMES=1000 python tools/scan_probe.py 2>&1 | tail -1 >> $L
C=4 python tools/fir_probe.py 2>&1 | tail -1 >> $L
C=4096 python tools/fir_probe.py 2>&1 | tail -1 >> $L
cat $L
