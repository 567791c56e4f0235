#!/bin/bash
# gpu_retry.sh <log> <timeout_s> <command...>: gpurun with retries while the pod's GPU slots are busy (exit 3 = nothing charged)
log=$1; shift; to=$1; shift
for i in $(seq 1 20); do
    /usr/local/graft/bin/gpurun --timeout "$to" -- "$@" > "$log" 2>&1
    rc=$?
    if ! grep -q "status=transient" "$log"; then break; fi
    sleep 90
done
echo "gpu_retry: finished (gpurun rc=$rc, attempt $i)" >> "$log"
