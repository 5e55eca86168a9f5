#!/bin/bash
cd "$(dirname "$0")/.."
L=gpurun_out/fir_ab.log
: > $L
run() { echo "## $*" >> $L; env "$@" timeout 600 python tools/fir_probe.py 2>&1 | tail -1 >> $L; }
for C in 256 1024; do
run C=$C STEPS=10
run C=$C STEPS=10 RCF_FIR_MFMA_PARTS=1
run C=$C STEPS=10 RCF_FIR_MFMA_PARTS=2
run C=$C STEPS=10 RCF_FIR_MFMA_PARTS=3
run C=$C STEPS=10 RCF_FIR_MFMA_PARTS=3 RCF_FIR_MFMA_NT=2
run C=$C STEPS=10 RCF_FIR_MFMA_PARTS=6
done
run C=64 STEPS=10
run C=4096 STEPS=5
cat $L
