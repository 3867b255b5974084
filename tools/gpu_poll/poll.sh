#!/bin/bash
# Retries gpurun until a call is accepted, then runs the round's whole GPU session from build/head (see refresh_head.sh).
# usage: tools/gpu_poll/poll.sh [interval_seconds]    log: /tmp/i2s_gpu_poll.log
I=${1:-480}
while true; do
  /usr/local/graft/bin/gpurun --timeout 4200 -- 'cd build/head && export GRAFT_REPO_ROOT=$PWD && echo "HEAD $(cat HEAD_SHA)" && bash tools/r06_gpu_session.sh; mkdir -p ../../gpurun_out && cp -r gpurun_out/* ../../gpurun_out/' > /tmp/i2s_gpu_poll_last.log 2>&1
  if ! grep -q "status=refused\|status=busy\|status=nobox" /tmp/i2s_gpu_poll_last.log; then echo "$(date) ACCEPTED" >> /tmp/i2s_gpu_poll.log; cat /tmp/i2s_gpu_poll_last.log >> /tmp/i2s_gpu_poll.log; exit 0; fi
  echo "$(date) $(grep -o 'status=[a-z]*' /tmp/i2s_gpu_poll_last.log | head -1)" >> /tmp/i2s_gpu_poll.log
  sleep $I
done
