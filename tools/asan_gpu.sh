#!/bin/bash
# The whole C ABI host layer (rcf_handle / rcf_plan / rcf_launch / rcf_chan / rcf_group ...: mutexes, deferred frees, slab pools, double-buffered arenas, bank-matrix
# cache) under AddressSanitizer on a GPU box, over the whole GPU suite (every test but the one that loads librccl).
#   make -C radiocapture-rf_amd/csrc asan      (here: the .so travels with the snapshot)
#   gpurun -- tools/asan_gpu.sh                -> gpurun_out/asan_gpu.txt
cd "$(dirname "$0")/.."
make -C radiocapture-rf_amd/csrc asan -j8 -s > /dev/null 2>&1     # never run a stale host layer (it must export every symbol native.py binds)
RT=$(gcc -print-file-name=libasan.so)
OUT=gpurun_out/asan_gpu.txt
mkdir -p gpurun_out
{
  echo "# ASan runtime: $RT"
  RCF_LIBRCF=$PWD/radiocapture-rf_amd/rcf/librcf_asan.so LD_PRELOAD=$RT \
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 \
  timeout 2000 python -m pytest tests -m gpu -x -q -k "not rccl_world1 and not 32_front_ends_at_20_msps" 2>&1 | tail -25
  # (librccl's own dlopen()s do not survive the preloaded ASan runtime; the 32 x 20 Msps real-time test asserts that no block
  # is late, which the instrumented host layer -- several times slower at planning a group block -- cannot promise)
  echo "# exit: $?"
} > $OUT 2>&1
cat $OUT
