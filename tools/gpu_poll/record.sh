#!/bin/bash
# Appends / refreshes the poller's summary at the end of profiles/r06_unverified_on_gpu.txt (everything after the marker line is replaced).
cd /root/repo
F=profiles/r06_unverified_on_gpu.txt
L=/tmp/i2s_gpu_poll.log
sed -i '/^== poller log ==/,$d' $F
{
  echo "== poller log =="
  echo "calls refused so far: $(grep -c 'status=refused' $L) (first $(head -1 $L | cut -d' ' -f1-5), last $(tail -1 $L | cut -d' ' -f1-5)); accepted: $(grep -c ACCEPTED $L)"
  echo "interval 8 min until 02:45 UTC, 3 min since; every call: status=refused, charged 0.0 s, \"closed from outside the build\""
} >> $F
