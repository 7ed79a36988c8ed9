#!/bin/bash
# gpurun with retries while no slot is free (exit code 3 = nothing charged).  usage: tools/gpurun_retry.sh <timeout> '<command>'
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
