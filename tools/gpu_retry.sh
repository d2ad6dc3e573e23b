#!/bin/bash
# usage: tools/gpu_retry.sh <logfile> <timeout> '<command>'  -- retries while the pod's GPU slots are busy.
# Exit status: gpurun's own status of the last attempt (a failed GPU command is reported, not swallowed);
# 75 (EX_TEMPFAIL) when every attempt was refused as transient.
log=$1; to=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1
  rc=$?
  if ! grep -q "status=transient" $log && [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
echo "gpu_retry: 40 attempts, still transient" >> $log
exit 75
