#!/bin/bash
# tools/gpu.sh <tag> <timeout-seconds> '<command>' — one gpurun call whose log lands in gpurun_out/<tag>/gpurun.log; retried while
# every GPU slot of the pod is busy (exit code 3: nothing is charged)
tag=$1; to=$2; shift 2
mkdir -p gpurun_out/$tag
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $to -- "mkdir -p gpurun_out/$tag; $*" > gpurun_out/$tag/gpurun.log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then break; fi
  sleep 45
done
echo "rc=$rc" >> gpurun_out/$tag/gpurun.log
