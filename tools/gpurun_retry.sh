#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3: nothing charged).  Usage: gpurun_retry.sh TIMEOUT 'command'
T="$1"; shift
git -C "$(dirname "$0")/.." rev-parse --short HEAD > "$(dirname "$0")/../.gpurun_commit" 2>/dev/null
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
