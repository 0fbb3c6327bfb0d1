#!/bin/bash
# usage: tools/_gpu_retry.sh <timeout> <script>   -- retries gpurun while the pod's GPU slots are busy (rc 3)
for i in $(seq 1 15); do
  /usr/local/graft/bin/gpurun --timeout $1 -- "bash $2" > /tmp/gpurun_last.txt 2>&1
  if ! grep -q "status=transient" /tmp/gpurun_last.txt; then break; fi
  sleep 90
done
tail -70 /tmp/gpurun_last.txt
