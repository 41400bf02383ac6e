#!/bin/bash
# gpurun wrapper for this session: waits for a free GPU slot (exit code 3 = nothing ran, nothing charged), never
# re-runs a command that started.  usage: tools/gpu.sh <tag> <timeout_s> '<command>'   -> gpurun_out/<tag>_call.log
TAG=$1; TMO=$2; CMD=$3
mkdir -p /root/repo/gpurun_out
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $TMO -- "$CMD" > /root/repo/gpurun_out/${TAG}_call.log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then tail -4 /root/repo/gpurun_out/${TAG}_call.log; exit $rc; fi
  sleep 90
done
echo "no GPU slot after 20 tries"; exit 3
