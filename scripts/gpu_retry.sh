#!/bin/bash
# gpurun with retries ONLY when no slot/box was free (exit 3: nothing ran, nothing charged).  usage: gpu_retry.sh <timeout> '<command>'
T=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
