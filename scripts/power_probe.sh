#!/bin/bash
# samples rocm-smi power / sclk while ab_probe runs with a given library:  scripts/power_probe.sh lib.so
cd "$(dirname "$0")/.."
REPS=${REPS:-12} GSP_LIB_PATH=$PWD/$1 timeout 200 python scripts/ab_probe.py > /tmp/ab_$$.txt 2>&1 &
pid=$!
sleep 3
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '; echo
  sleep 0.25
done
cat /tmp/ab_$$.txt | tail -1
