#!/bin/bash
# scripts/grun.sh with retries while every GPU slot of the pod is busy (status=transient, nothing charged).
# Exit code: gpurun's own (the command's verdict) once a call went through; 3 when every retry was transient.
for i in 1 2 3 4 5 6 7 8; do
  out=$("$(dirname "$0")/grun.sh" "$1" "$2" 2>&1)
  rc=$?
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out"; exit $rc
done
echo "$out"; exit 3
