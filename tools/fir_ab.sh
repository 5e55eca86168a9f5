#!/bin/bash
cd "$(dirname "$0")/.."
L=gpurun_out/fir_ab.log
: > $L
run() { echo "## $*" >> $L; env "$@" timeout 600 python tools/fir_probe.py 2>&1 | tail -1 >> $L; }
for C in 256 1024 4096 16384; do run C=$C STEPS=10; done
run C=64 CR=6250
run C=512 CR=6250
run C=1024 FS=2400000 BLOCK=1048576
cat $L
