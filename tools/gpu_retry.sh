#!/bin/bash
# usage: tools/gpu_retry.sh <logfile> <timeout> '<command>'  -- retries while the pod's GPU slots are busy
log=$1; to=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1
  if ! grep -q "status=transient" $log; then exit 0; fi
  sleep 45
done
