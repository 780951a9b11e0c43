#!/bin/bash
# gpurun_retry.sh <out file> <timeout> <command> -- gpurun with retries while the pod's GPU slots are busy (exit code 3: nothing charged)
OUT=$1; TO=$2; shift 2
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $TO -- "$@" > $OUT 2>&1; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
