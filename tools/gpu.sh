#!/bin/bash
# gpurun with a polite retry while every GPU slot of the pod is busy (exit code 3: nothing ran, nothing was charged).
# usage: tools/gpu.sh <timeout-seconds> '<command>'
t=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
