#!/bin/bash
# A/B of librcf variants on the filterbank probe:  tools/ab_pfb.sh "<nb list>" "<variant tags; '-' = librcf.so>" [env...]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/ab_pfb.txt; : > $O
L=$PWD/radiocapture-rf_amd/rcf
NBS=$1; VARS=$2; shift 2
for rep in 1 2; do
for nb in $NBS; do
  for v in $VARS; do
    lib=$L/librcf_$v.so; [ "$v" = "-" ] && lib=$L/librcf.so
    printf "%-8s " $v >> $O
    env "$@" RCF_LIBRCF=$lib NB=$nb WARM=${WARM:-200} STEPS=${STEPS:-300} python tools/pfb_probe.py >> $O 2>&1
  done
done
done
cat $O
