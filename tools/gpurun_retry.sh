#!/bin/bash
# gpurun with a retry ONLY for exit code 3 (no box / slot free: nothing ran, nothing charged).  Any other result is final.
# usage: tools/gpurun_retry.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
