#!/bin/bash
# queue helper for this container: gpurun returns 3 when no GPU slot is free (nothing charged) -- wait and ask again.
# usage: tools/gq.sh <timeout-seconds> '<command>'    (the command itself is never retried once it has run)
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
