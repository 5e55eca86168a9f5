#!/bin/bash
# build a variant of librcf.so with extra -D flags on ONE kernel file:  tools/variant_lib.sh <tag> <file.hip> <flags...>
# -> radiocapture-rf_amd/rcf/librcf_<tag>.so (git-ignored; travels with gpurun); use with RCF_LIBRCF=...
set -e
cd "$(dirname "$0")/../radiocapture-rf_amd/csrc"
TAG=$1; F=$2; shift 2
make -s -j8
base=${F%.hip}
EXTRA=""; case $F in pfb.hip|scan.hip) EXTRA="-fno-slp-vectorize";; pfb5.hip|tapfin.hip) EXTRA="-fno-slp-vectorize -ffp-contract=off";; fir.hip|peaks.hip|audio.hip) EXTRA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form $EXTRA "$@" -c $F -o build/${base}_$TAG.o
OBJS=$(ls build/*.o | grep -v "_[a-zA-Z0-9]*\.o$" | grep -v "build/$base.o"; ls build/rcf_*.o 2>/dev/null)
OBJS=$(echo $OBJS | tr ' ' '\n' | sort -u | grep -v "build/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../rcf/librcf_$TAG.so $OBJS build/${base}_$TAG.o -ldl
echo built ../rcf/librcf_$TAG.so
