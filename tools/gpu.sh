#!/bin/bash
# gpurun with retries while no slot / box is free (exit code 3):  tools/gpu.sh <timeout_s> '<command>'
T=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  grep -q '"status": *"transient"' gpurun_out/.last_call.json 2>/dev/null || true
  sleep 60
done
exit 3
