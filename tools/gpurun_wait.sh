#!/bin/bash
# gpurun, retried only while the pool reports "no box or slot free right now" (exit code 3: nothing ran, nothing charged).
# Any other outcome — including a failed or killed command — is returned as is, never retried.
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
