#!/bin/bash
# usage: tools/gpu.sh <timeout-seconds> '<command>'
# Runs one gpurun call.  Only "no box free right now" (exit code 3: nothing ran, nothing charged) is
# retried, after a pause; any result of the command itself is final.
limit=$1
shift
for attempt in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout "$limit" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  echo "[gpu.sh] no box free (attempt $attempt); waiting 150 s" >&2
  sleep 150
done
exit 3
